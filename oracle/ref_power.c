/*
 * ref_power.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Dumps the POWER samples the reference's real decodeiq (air.c:54-101, compiled
 * from /root/reference by oracle/Makefile, see ref_chain_tail.c) produces for a
 * uint16 capture.  This program sits on the far side of the reference's own seam
 * `extern int deqframe(const float *ampbuff, const int len)` (adsbdec.h:5): its
 * deqframe is a recorder that consumes len-DECOFFSET samples (a value the real one
 * may return too), so every power sample passes through it exactly once; what is
 * still in ampbuff at EOF is read through ref_ampbuff()/ref_aidx().
 *
 * usage: ref_power [-c chunk_samples] in.u16 out.f32
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

extern void ref_decodeiq(const unsigned short *r, int len);
extern uint32_t ref_aidx(void);
extern const float *ref_ampbuff(void);

static FILE *out;

int deqframe(const float *ampbuff, const int len)
{
    int keep = 1200; /* DECOFFSET */
    fwrite(ampbuff, sizeof(float), (size_t)(len - keep), out);
    return len - keep;
}

int main(int argc, char **argv)
{
    size_t chunk = 1024 * 1024; /* IQBUFFSZ, air.c:218 */
    int argi = 1;
    if (argi + 1 < argc && strcmp(argv[argi], "-c") == 0) {
        chunk = strtoull(argv[argi + 1], NULL, 0);
        argi += 2;
    }
    if (argi + 2 != argc || chunk == 0) {
        fprintf(stderr, "usage: ref_power [-c chunk_samples] in.u16 out.f32\n");
        return 2;
    }
    FILE *in = fopen(argv[argi], "rb");
    out = fopen(argv[argi + 1], "wb");
    if (!in || !out) {
        perror("open");
        return 2;
    }
    /* like fileInput (air.c:230-239): one buffer, reused; +4 so a ragged tail
     * over-reads inside the allocation (the reference reads stale bytes there) */
    unsigned short *buf = calloc(chunk + 4, sizeof *buf);
    size_t n;
    while ((n = fread(buf, sizeof *buf, chunk, in)) > 0)
        ref_decodeiq(buf, (int)n);
    fwrite(ref_ampbuff(), sizeof(float), ref_aidx(), out);
    fclose(out);
    fclose(in);
    free(buf);
    return 0;
}
