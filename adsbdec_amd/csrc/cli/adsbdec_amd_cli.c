/*
 * adsbdec_amd_cli.c -- host program in C for the offline "-f" path, calling the
 * HIP library through the C-ABI of include/adsbdec_amd.h.
 *
 * Mirrors the reference's command line for this path (main.c:60-89):
 *     -f filename   input file (air.c:217-246 fileInput)
 *     -a            also decode DF18 (sets `df`, main.c:76-78 / demod.c:26)
 *     -m            AVR-MLAT output (outformat 1, main.c:79-81)
 *     -b            Beast binary output (outformat 2, main.c:82-84)
 *     -g n          accepted and ignored (gain only matters for the live radio)
 *     -x            EXTENSION, not in the reference: repair single-bit errors in DF17/18
 *                   frames (the reference's -e flag is parsed but does nothing and exits
 *                   with the usage text, main.c:40,60,85-87; that behaviour is kept for -e)
 * -s / -l (TCP sinks), the live Airspy input and anything else print the usage
 * text and exit 1, like the reference's default: branch (main.c:85-87).
 *
 * Differences from the reference, on purpose (DESIGN.md "CLI"):
 *   - every accepted frame is written: the reference drops frames still queued
 *     when the reader thread hits EOF;
 *   - Beast to stdout is written with its real length (the reference uses strlen()
 *     on a binary buffer).
 * The stderr statistics table has the reference's format (valid.c:84-100).
 */
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>

#include "adsbdec_amd.h"

#define READ_SAMPLES (16u * 1024u * 1024u) /* 32 MiB per read(); the reference reads 2 MiB */

static void usage(void)
{
    printf("adsbdec_amd : MI355X offline ADS-B decoder (adsbdec -f compatible)\n\n");
    printf("usage : adsbdec_amd_cli [-a] [-m] [-b] -f filename\n\n");
    printf("\t-a : decode DF18 too\n");
    printf("\t-m : output avrmlat format (ie : with 12Mhz timestamp)\n");
    printf("\t-b : output binary beast format\n");
    printf("\t-x : (extension) repair 1-bit CRC errors in DF17/18 frames\n");
    printf("\t-f : input from filename (raw 16 bits real, 12-bit ADC code centred on 2048)\n");
}

static int flush_frames(adsb_decoder *dec, int outformat)
{
    const adsb_frame *fr; /* the handle's own queue: formatted where it lies */
    char pkt[256];
    const long n = adsb_take(dec, &fr);
    for (long i = 0; i < n; i++) {
        int len = adsb_format_frame(&fr[i], outformat, pkt);
        if (fwrite(pkt, 1, (size_t)len, stdout) != (size_t)len)
            return -1;
    }
    return n < 0 ? -1 : 0;
}

int main(int argc, char **argv)
{
    const char *filename = NULL;
    int outformat = 0, df18 = 0, fix1 = 0, c;

    while ((c = getopt(argc, argv, "f:g:ambx")) != EOF) {
        switch (c) {
        case 'f':
            filename = optarg;
            break;
        case 'g':
            break;
        case 'a':
            df18 = 1;
            break;
        case 'm':
            outformat = 1;
            break;
        case 'b':
            outformat = 2;
            break;
        case 'x':
            fix1 = 1;
            break;
        default:
            usage();
            return 1;
        }
    }
    if (!filename) {
        usage();
        return 1;
    }

    adsb_config cfg;
    adsb_config_default(&cfg);
    cfg.df18 = df18;
    cfg.fix_1bit = fix1;
    cfg.collect_stats = 1; /* the reference always prints Try/Ok */
    adsb_decoder *dec = adsb_create(&cfg);
    if (!dec) {
        fprintf(stderr, "adsb_create() failed: %s\n", adsb_last_error(NULL));
        return 255; /* runOutput() == -1 -> exit status 255 (main.c:101-105) */
    }

    int rc = 0;
    int fd = open(filename, O_RDONLY);
    if (fd >= 0) { /* an unopenable file ends the run silently (air.c:225-228) */
        /* page-locked, so that adsb_push() is one DMA (air.c:230 uses malloc) */
        uint16_t *buf = (uint16_t *)adsb_host_alloc((size_t)READ_SAMPLES * sizeof(uint16_t));
        if (!buf) {
            fprintf(stderr, "adsb_host_alloc() failed\n");
            adsb_destroy(dec);
            return 255;
        }
        size_t have = 0; /* bytes carried when read() returns an odd count */
        for (;;) {
            ssize_t n = read(fd, (char *)buf + have, (size_t)READ_SAMPLES * 2 - have);
            if (n <= 0)
                break;
            size_t bytes = have + (size_t)n;
            if (adsb_push(dec, buf, bytes / 2) != 0) {
                fprintf(stderr, "adsb_push() failed: %s\n", adsb_last_error(dec));
                rc = 255;
                break;
            }
            have = bytes & 1;
            if (have)
                ((char *)buf)[0] = ((char *)buf)[bytes - 1];
            if (flush_frames(dec, outformat) != 0)
                break;
        }
        adsb_host_free(buf);
        close(fd);
        if (rc == 0 && adsb_finish(dec) != 0) {
            fprintf(stderr, "adsb_finish() failed: %s\n", adsb_last_error(dec));
            rc = 255;
        }
        flush_frames(dec, outformat);
        fflush(stdout);
    }

    adsb_stats st;
    if (adsb_get_stats(dec, &st) == 0) { /* valid.c:84-100 */
        unsigned long long tot = st.ok[0] + st.ok[1] + st.ok[2];
        fprintf(stderr, "\t%10d\t%10d\t%10d\n", 11, 17, 18);
        fprintf(stderr, "Try :\t%10llu\t%10llu\t%10llu\n", (unsigned long long)st.try_[0],
                (unsigned long long)st.try_[1], (unsigned long long)st.try_[2]);
        fprintf(stderr, "Ok :\t%10llu\t%10llu\t%10llu\n", (unsigned long long)st.ok[0],
                (unsigned long long)st.ok[1], (unsigned long long)st.ok[2]);
        fprintf(stderr, "Total :\t%10llu\n", tot); /* tot_fi is uninitialised there */
    }
    adsb_destroy(dec);
    return rc;
}
