/*
 * dropin_decodeiq.c -- TEST INFRASTRUCTURE ONLY.
 *
 * INTEGRATION.md's patch as compiled code: the body a maintainer would give decodeiq (air.c:54-101) so that the
 * reference's `-f` path runs on libadsbdec_amd, linked with ref_harness.c (the fileInput-shaped read loop and the
 * netout() collector that renders every frame with the REAL formatpkt of output.c) into
 * oracle/_ref/ref_adsbdec_dropin.  demod.c, valid.c and crc.h are NOT linked into that binary: what they did is
 * what the library does now.  tests/test_gpu_dropin.py runs it beside oracle/_ref/ref_adsbdec (the unpatched chain)
 * on the same capture files and compares stdout and the Try/Ok table byte for byte.
 *
 * The product never links this file; it only shows (and tests) how the product is bound.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../include/adsbdec_amd.h"

int df = 0;                                              /* demod.c:26 lives here now: set by -a (main.c:76-78) */
extern void netout(const uint8_t *frame, const int len,
                   const uint64_t ts, const uint32_t pw); /* output.c:159 (ref_harness.c's collector in this build) */

static adsb_decoder *gpu;

static void gpu_flush(void)                              /* hands accepted frames to the existing queue */
{
    adsb_frame fr[64];
    long n;
    while ((n = adsb_drain(gpu, fr, 64)) > 0)
        for (long i = 0; i < n; i++)
            netout(fr[i].frame, fr[i].len, fr[i].ts, fr[i].pw); /* same order as valid.c:54,76 */
}

static void decodeiq(const unsigned short *r, const int len) /* replaces air.c:54-101 */
{
    if (!gpu) {
        adsb_config cfg;
        adsb_config_default(&cfg);
        cfg.df18 = df;
        cfg.collect_stats = 1;                           /* keeps print_stats() meaningful */
        if (getenv("DROPIN_PUSH_OVERLAP"))
            cfg.push_overlap = 1;                        /* the one-more-line variant */
        gpu = adsb_create(&cfg);
        if (!gpu) {
            fprintf(stderr, "adsb_create() failed: %s\n", adsb_last_error(NULL));
            exit(1);
        }
    }
    if (adsb_push(gpu, r, (size_t)len) != 0) {           /* borrowed buffer, like the original */
        fprintf(stderr, "adsb_push() failed: %s\n", adsb_last_error(gpu));
        exit(1);
    }
    gpu_flush();
}

/* the harness's names for the two call sites */
void ref_decodeiq(const unsigned short *r, int len) { decodeiq(r, len); }

void dropin_eof(void)                                    /* fileInput at EOF, before handlerExit(0) (air.c:241-244) */
{
    if (!gpu)
        return;
    adsb_finish(gpu);
    gpu_flush();
}

void print_stats(void)                                   /* valid.c:84-100 reading adsb_get_stats instead of stat_try/stat_ok */
{
    adsb_stats st = {{0, 0, 0}, {0, 0, 0}, 0};
    if (gpu)
        adsb_get_stats(gpu, &st);
    fprintf(stderr, "\t%10d\t%10d\t%10d\n", 11, 17, 18);
    fprintf(stderr, "Try :\t%10d\t%10d\t%10d\n", (int)st.try_[0], (int)st.try_[1], (int)st.try_[2]);
    fprintf(stderr, "Ok :\t%10d\t%10d\t%10d\n", (int)st.ok[0], (int)st.ok[1], (int)st.ok[2]);
    fprintf(stderr, "Total :\t%10d\n", (int)(st.ok[0] + st.ok[1] + st.ok[2]));
}
