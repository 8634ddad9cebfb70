/*
 * ref_harness.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Driver around the REAL reference code (compiled by oracle/Makefile from the
 * sources where they lie under /root/reference, outputs only in oracle/_ref/):
 *     air.c:29-101 -> fbuff/fidx, dsfilter, ampbuff/aidx, decodeiq  (the slice that
 *                 uses no libairspy symbol; see ref_chain_tail.c for how it is built)
 *     demod.c  -> deqframe/getdf/getabyte        (demod.c:31-144)
 *     valid.c  -> validShort/validLong/print_stats + crc.h   (valid.c:39-101)
 *     output.c -> formatpkt                       (output.c:204-262), linked as
 *                 _ref/libref_format.so with everything else garbage-collected.
 * What this file adds is only what stands on the OUTER side of the reference's own
 * seams: a read loop shaped like fileInput (air.c:217-246: one reused buffer,
 * IQBUFFSZ = 1 Mi samples per call of decodeiq) and the egress callback netout()
 * (valid.c:26) as a collector, exactly like output.c's writer.
 *
 * usage: ref_adsbdec [-a] [-c chunk_samples] capture.u16
 *            the real chain decodeiq -> deqframe -> validShort/Long -> formatpkt
 *        ref_adsbdec [-a] -p power.f32
 *            POWER samples (float32, one per 10 MS/s sample) fed through a
 *            restatement of air.c:94-99's accumulate/carry into the real deqframe:
 *            used to fuzz getdf/getabyte on synthetic power, not for pinning
 *   -> stdout: one line per accepted frame
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
extern void ref_decodeiq(const unsigned short *r, int len); /* ref_chain_tail.c -> air.c:54 */

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

#ifdef DROPIN /* oracle/_ref/ref_adsbdec_dropin: decodeiq is INTEGRATION.md's patch (dropin_decodeiq.c), no demod.c / valid.c */
extern void dropin_eof(void);
static int run_power(const char *path)
{
    (void)path;
    fprintf(stderr, "-p needs the reference's deqframe: not in the drop-in build\n");
    return 2;
}
#else
static int run_power(const char *path)
{
    FILE *f = fopen(path, "rb");
    if (!f) {
        perror(path);
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
    return 0;
}
#endif

/* fileInput (air.c:217-246): read() into one reused buffer, decodeiq(iqbuff, n/2).
 * The buffer has 4 spare samples so that a ragged last read over-reads inside the
 * allocation (the reference reads whatever the buffer held there). */
static int run_capture(const char *path, size_t chunk)
{
    FILE *f = fopen(path, "rb");
    if (!f) {
        perror(path);
        return 2;
    }
    unsigned short *iqbuff = calloc(chunk + 4, sizeof *iqbuff);
    size_t n;
    while ((n = fread(iqbuff, sizeof *iqbuff, chunk, f)) > 0)
        ref_decodeiq(iqbuff, (int)n);
#ifdef DROPIN
    dropin_eof(); /* the patch's one line at EOF (air.c:241-244) */
#endif
    free(iqbuff);
    fclose(f);
    return 0;
}

int main(int argc, char **argv)
{
    int argi = 1, power = 0;
    size_t chunk = 1024 * 1024; /* IQBUFFSZ, air.c:218 */
    for (; argi < argc && argv[argi][0] == '-'; argi++) {
        if (strcmp(argv[argi], "-a") == 0)
            df = 1; /* main.c:76-78 */
        else if (strcmp(argv[argi], "-p") == 0)
            power = 1;
        else if (strcmp(argv[argi], "-c") == 0 && argi + 1 < argc)
            chunk = strtoull(argv[++argi], NULL, 0);
        else
            break;
    }
    if (argi + 1 != argc || chunk == 0) {
        fprintf(stderr, "usage: ref_adsbdec [-a] [-c chunk_samples] capture.u16 | [-a] -p power.f32\n");
        return 2;
    }
    int rc = power ? run_power(argv[argi]) : run_capture(argv[argi], chunk);
    if (rc)
        return rc;
    fflush(stdout);
    print_stats();
    return 0;
}
