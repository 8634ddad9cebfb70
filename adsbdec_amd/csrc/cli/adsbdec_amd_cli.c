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
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include "adsbdec_amd.h"

/* fileInput (air.c:217-246) reads 2 MiB at a time into one buffer and decodes it before the
 * next read().  Here a reader thread fills a ring of 32 MiB buffers from the moment the
 * program starts -- so the file is being read while the GPU runtime initialises (~0.2 s) --
 * and the main thread hands each filled buffer to adsb_push_async(), which overlaps the
 * host-to-device copy of one buffer with the scan of the previous one. */
#define BUF_SAMPLES (16u * 1024u * 1024u)
#define RING_MAX_BYTES (1024ull * 1024ull * 1024ull)

typedef struct {
    uint16_t *buf;
    size_t bytes; /* valid bytes once filled */
    int filled;
    int registered;
    int ready;    /* filled, and page-locked if that was asked for: the main thread may push it */
} ring_slot;

typedef struct {
    int fd, nbuf, failed, use_register;
    double t_reg;
    ring_slot *slot;
    pthread_mutex_t mu;
    pthread_cond_t cv;
} ring;

static void *reader_main(void *arg)
{
    ring *r = (ring *)arg;
    const size_t cap = (size_t)BUF_SAMPLES * 2;
    for (int k = 0;; k++) {
        ring_slot *s = &r->slot[k % r->nbuf];
        pthread_mutex_lock(&r->mu);
        while (s->filled)
            pthread_cond_wait(&r->cv, &r->mu);
        pthread_mutex_unlock(&r->mu);
        if (!s->buf) {
            /* 2 MiB-aligned and marked for transparent huge pages: first-touching 32 MiB then takes 16
             * page faults instead of 8192 -- the faults of a 4 KiB-page buffer hold the address-space
             * lock that the GPU runtime's start-up (hundreds of mmaps) needs at the same moment, and
             * were measured to slow it down by more than the overlap gained */
            if (posix_memalign((void **)&s->buf, 2u << 20, cap) != 0) {
                s->buf = NULL;
                r->failed = 1;
            } else {
                madvise(s->buf, cap, MADV_HUGEPAGE);
            }
        }
        size_t got = 0;
        while (s->buf && got < cap) { /* fill the buffer: only the last one of a file is short */
            ssize_t n = read(r->fd, (char *)s->buf + got, cap - got);
            if (n <= 0)
                break;
            got += (size_t)n;
        }
        pthread_mutex_lock(&r->mu);
        s->bytes = got;
        s->filled = 1;
        pthread_cond_broadcast(&r->cv);
        pthread_mutex_unlock(&r->mu);
        if (got < cap)
            return NULL; /* end of file (or an error: the run ends there, air.c:236-237) */
    }
}

static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

/* Started once the GPU runtime is up: page-locks the filled buffers in ring order (each one
 * once), so that buffer k+1 is being locked while buffer k is pushed. */
static void *locker_main(void *arg)
{
    ring *r = (ring *)arg;
    for (int k = 0;; k++) {
        ring_slot *s = &r->slot[k % r->nbuf];
        pthread_mutex_lock(&r->mu);
        while (!s->filled || s->ready)
            pthread_cond_wait(&r->cv, &r->mu);
        pthread_mutex_unlock(&r->mu);
        const size_t bytes = s->bytes;
        if (r->use_register && !s->registered && s->buf && bytes >= 2) {
            const double t0 = now_ms();
            s->registered = adsb_host_register(s->buf, (size_t)BUF_SAMPLES * 2) == 0;
            r->t_reg += now_ms() - t0;
        }
        pthread_mutex_lock(&r->mu);
        s->ready = 1;
        pthread_cond_broadcast(&r->cv);
        pthread_mutex_unlock(&r->mu);
        if (bytes < (size_t)BUF_SAMPLES * 2)
            return NULL;
    }
}

static void usage(void)
{
    printf("adsbdec_amd : MI355X offline ADS-B decoder (adsbdec -f compatible)\n\n");
    printf("usage : adsbdec_amd_cli [-a] [-m] [-b] -f filename\n\n");
    printf("\t-a : decode DF18 too\n");
    printf("\t-m : output avrmlat format (ie : with 12Mhz timestamp)\n");
    printf("\t-b : output binary beast format\n");
    printf("\t-x : (extension) repair 1-bit CRC errors in DF17/18 frames\n");
    printf("\t-f : input from filename (raw 16 bits real: uint16 carrying the 12-bit ADC code centred on 2048;\n");
    printf("\t     bit-identical to adsbdec for codes 0..4095, see adsbdec_amd.h for the wider domain)\n");
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

    const int timing = getenv("ADSB_CLI_TIMING") != NULL;
    const int use_register = !(getenv("ADSB_CLI_REGISTER") && atoi(getenv("ADSB_CLI_REGISTER")) == 0);
    const double t_start = now_ms();

    /* start reading before anything touches the GPU */
    ring rg;
    memset(&rg, 0, sizeof rg);
    pthread_t reader;
    int have_reader = 0;
    rg.fd = open(filename, O_RDONLY);
    if (rg.fd >= 0) { /* an unopenable file ends the run silently (air.c:225-228) */
        struct stat sb;
        unsigned long long size = (fstat(rg.fd, &sb) == 0 && sb.st_size > 0) ? (unsigned long long)sb.st_size : 0;
        if (size > RING_MAX_BYTES || size == 0)
            size = RING_MAX_BYTES; /* pipes and huge files: a bounded ring, the reader waits for free buffers */
        rg.nbuf = (int)(size / ((unsigned long long)BUF_SAMPLES * 2)) + 2;
        if (rg.nbuf < 3)
            rg.nbuf = 3;
        rg.slot = (ring_slot *)calloc((size_t)rg.nbuf, sizeof *rg.slot);
        pthread_mutex_init(&rg.mu, NULL);
        pthread_cond_init(&rg.cv, NULL);
        have_reader = rg.slot && pthread_create(&reader, NULL, reader_main, &rg) == 0;
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
    const double t_init = now_ms();

    int rc = 0;
    pthread_t locker;
    rg.use_register = use_register;
    if (have_reader && pthread_create(&locker, NULL, locker_main, &rg) != 0)
        have_reader = 0;
    if (have_reader) {
        int prev = -1;
        for (int k = 0;; k++) {
            ring_slot *s = &rg.slot[k % rg.nbuf];
            pthread_mutex_lock(&rg.mu);
            while (!s->ready)
                pthread_cond_wait(&rg.cv, &rg.mu);
            const int reader_failed = rg.failed; /* (written under the mutex by the reader) */
            pthread_mutex_unlock(&rg.mu);
            if (reader_failed) {
                fprintf(stderr, "out of memory for the read buffers\n");
                rc = 255;
                break;
            }
            const size_t bytes = s->bytes;
            if (bytes >= 2) {
                /* a trailing odd byte is dropped, like decodeiq(iqbuff, n / 2) (air.c:239) */
                const int prc = s->registered ? adsb_push_async(dec, s->buf, bytes / 2) : adsb_push(dec, s->buf, bytes / 2);
                if (prc != 0) {
                    fprintf(stderr, "adsb_push() failed: %s\n", adsb_last_error(dec));
                    rc = 255;
                    break;
                }
                if (flush_frames(dec, outformat) != 0) { /* stdout is gone (EPIPE, disk full): stop, and say so */
                    rc = 1;
                    break;
                }
            }
            if (prev >= 0) { /* the buffer of the previous push is free again (adsb_push_async's contract) */
                pthread_mutex_lock(&rg.mu);
                rg.slot[prev].ready = 0;
                rg.slot[prev].filled = 0;
                pthread_cond_broadcast(&rg.cv);
                pthread_mutex_unlock(&rg.mu);
            }
            prev = k % rg.nbuf;
            if (bytes < (size_t)BUF_SAMPLES * 2)
                break; /* that was the last buffer */
        }
        if (rc == 0 && adsb_finish(dec) != 0) {
            fprintf(stderr, "adsb_finish() failed: %s\n", adsb_last_error(dec));
            rc = 255;
        }
        flush_frames(dec, outformat);
        fflush(stdout);
        if (rc != 0) { /* let the reader run out: hand every buffer back */
            pthread_mutex_lock(&rg.mu);
            for (int i = 0; i < rg.nbuf; i++)
                rg.slot[i].ready = rg.slot[i].filled = 0;
            pthread_cond_broadcast(&rg.cv);
            pthread_mutex_unlock(&rg.mu);
            close(rg.fd); /* read() fails from here on */
        }
        if (rc == 0) { /* (after a failure the threads are left to the process exit) */
            pthread_join(reader, NULL);
            pthread_join(locker, NULL);
            close(rg.fd);
        }
    }
    const double t_done = now_ms();
    if (timing)
        fprintf(stderr, "timing: runtime init %.1f ms, decode %.1f ms (page-locking, on its own thread: %.1f ms), total %.1f ms\n",
                t_init - t_start, t_done - t_init, rg.t_reg, t_done - t_start);

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
    /* no adsb_destroy / unregister / free: the process ends here, and tearing the GPU runtime
     * down cleanly costs tens of milliseconds that an offline decode has no use for */
    fflush(stderr);
    _exit(rc);
}
