/*
 * ref_harness.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Driver around the REAL reference objects (compiled by oracle/Makefile from the
 * sources where they lie under /root/reference, outputs only in oracle/_ref/):
 *     demod.c  -> deqframe/getdf/getabyte        (demod.c:31-144)
 *     valid.c  -> validShort/validLong/print_stats + crc.h   (valid.c:39-101)
 *     output.c -> formatpkt                       (output.c:204-262), linked as
 *                 _ref/libref_format.so with everything else garbage-collected.
 * air.c is NOT built (needs <libairspy/airspy.h>, absent from this image; writing a
 * stand-in header is not allowed), so this driver takes POWER samples (float32,
 * one per 10 MS/s sample) and replays air.c:94-99's accumulate/carry around the
 * real deqframe.  It implements the egress seam netout() (valid.c:26) as a
 * collector: it is the consumer of that callback, exactly like output.c's writer.
 *
 * usage: ref_demod [-a] power.f32   -> stdout: one line per accepted frame
 *        "<ts> <pw> <len> <avr line> <mlat line> <beast hex>"; stderr: print_stats().
 */
#include <inttypes.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* reference globals / entry points */
extern int df;                                          /* demod.c:26 */
extern int deqframe(const float *ampbuff, const int len); /* adsbdec.h:5 */
extern void print_stats(void);                          /* valid.c:84 */
extern int outformat;                                   /* output.c:40 */

/* layout of output.c:45-52, needed to call the real formatpkt */
typedef struct ref_blk_s {
    uint8_t frame[112];
    int len;
    uint64_t ts;
    uint32_t pw;
    struct ref_blk_s *next;
} ref_blk;
extern int formatpkt(ref_blk *blk, char *pkt);          /* output.c:204 */

#define APBUFFSZ (8196 * 5)                             /* air.c:47 */

static void emit_hex(const char *p, int n)
{
    for (int i = 0; i < n; i++)
        printf("%02X", (unsigned char)p[i]);
}

/* egress seam: called by validShort/validLong on every accepted frame */
void netout(const uint8_t *frame, const int len, const uint64_t ts, const uint32_t pw)
{
    ref_blk blk;
    char pkt[256];
    memset(&blk, 0, sizeof blk);
    memcpy(blk.frame, frame, (size_t)len);
    blk.len = len;
    blk.ts = ts;
    blk.pw = pw;

    printf("%" PRIu64 " %u %d ", ts, pw, len);
    outformat = 0;
    int n = formatpkt(&blk, pkt);
    fwrite(pkt, 1, (size_t)n - 1, stdout); /* drop '\n' */
    putchar(' ');
    outformat = 1;
    n = formatpkt(&blk, pkt);
    fwrite(pkt, 1, (size_t)n - 1, stdout);
    putchar(' ');
    outformat = 2;
    n = formatpkt(&blk, pkt);
    emit_hex(pkt, n);
    putchar('\n');
}

int main(int argc, char **argv)
{
    int argi = 1;
    if (argi < argc && strcmp(argv[argi], "-a") == 0) {
        df = 1; /* main.c:76-78 */
        argi++;
    }
    if (argi >= argc) {
        fprintf(stderr, "usage: ref_demod [-a] power.f32\n");
        return 2;
    }
    FILE *f = fopen(argv[argi], "rb");
    if (!f) {
        perror(argv[argi]);
        return 2;
    }
    static float ampbuff[APBUFFSZ + 4]; /* air.c:49 */
    uint32_t aidx = 0;
    float pair[2];
    /* air.c:59-100 appends two power samples per loop pass, then tests aidx */
    while (fread(pair, sizeof(float), 2, f) == 2) {
        ampbuff[aidx++] = pair[0];
        ampbuff[aidx++] = pair[1];
        if (aidx >= APBUFFSZ) {
            int rlen = deqframe(ampbuff, (int)aidx);
            if ((uint32_t)rlen < aidx)
                memmove(ampbuff, &ampbuff[rlen], (aidx - (uint32_t)rlen) * sizeof(float));
            aidx -= (uint32_t)rlen;
        }
    }
    fclose(f);
    fflush(stdout);
    print_stats();
    return 0;
}
