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
 *     -d k          EXTENSION: decode on GPU k of the node (default: the first one)
 *     -G n | a,b,c  EXTENSION: shard the file over n GPUs (or over the GPUs listed; an ordinal may repeat) through
 *                   the library's multi-GPU driver (adsb_multi_decode_file): same bytes on stdout and stderr.
 *                   With several -f (one capture each) the captures are decoded side by side, one per GPU, and
 *                   capture k's packets go to <file k>.avr / .mlat / .beast instead of stdout.
 * The GPU runtime initialises every device it can see, which takes longer the more there are: before its first call
 * this program narrows ROCR_VISIBLE_DEVICES to the devices it is going to use (unless the variable is already set).
 *     -s addr[:port]  send the packets to a TCP peer instead of stdout (main.c:65-68; default port 30001)
 *     -l addr[:port]  listen, accept ONE peer and send the packets to it (main.c:69-72; default port 30002); sink.c
 * The live Airspy input and anything else print the usage text and exit 1, like the reference's default: branch
 * (main.c:85-87).
 *
 * Differences from the reference, on purpose (DESIGN.md "CLI"):
 *   - every accepted frame is written: the reference drops frames still queued
 *     when the reader thread hits EOF;
 *   - Beast to stdout is written with its real length (the reference uses strlen()
 *     on a binary buffer).
 * The stderr statistics table has the reference's format (valid.c:84-100).
 * Signals as in main.c:91-99: SIGPIPE ignored; SIGINT / SIGTERM / SIGQUIT stop the pushes, what has been decoded is
 * finished and written, the table is printed, exit status 0.
 */
#include <fcntl.h>
#include <errno.h>
#include <pthread.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include "adsbdec_amd.h"
#include "sink.h"

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
    printf("usage : adsbdec_amd_cli [-a] [-m] [-b] [-s addr[:port] | -l addr[:port]] [-d gpu | -G gpus] -f filename [-f filename ...]\n\n");
    printf("\t-a : decode DF18 too\n");
    printf("\t-m : output avrmlat format (ie : with 12Mhz timestamp)\n");
    printf("\t-b : output binary beast format\n");
    printf("\t-s addr[:port] : send ouput via TCP to addr:port (default port 30001)\n");
    printf("\t-l addr[:port] : listen to addr:port (default port 30002) and send ouput to the peer that connects\n");
    printf("\t-x : (extension) repair 1-bit CRC errors in DF17/18 frames\n");
    printf("\t-d k : (extension) use GPU k\n");
    printf("\t-G n | a,b,.. : (extension) shard the file over n GPUs / the GPUs listed; several -f: one capture per GPU,\n");
    printf("\t     packets of capture k written to <file k>.avr | .mlat | .beast\n");
    printf("\t-f : input from filename (raw 16 bits real: uint16 carrying the 12-bit ADC code centred on 2048;\n");
    printf("\t     bit-identical to adsbdec for codes 0..4095, see adsbdec_amd.h for the wider domain)\n");
}

static sink out_sink; /* stdout unless -s / -l */

/* main.c:91-99: SIGINT / SIGTERM / SIGQUIT end the run in an orderly way -- the reference's handlerExit sets do_exit, the
 * writer loop ends, main prints the Try/Ok table and returns runOutput()'s 0 -- and SIGPIPE is ignored, so that a closed
 * stdout or peer shows up as a failed write, not as death by signal.  Here: the handler sets a flag (no SA_RESTART: a
 * blocking accept / connect / read returns), the push loop stops at the next buffer, what has been decoded is finished,
 * written and counted, the table is printed, exit status 0. */
static volatile int stop_requested;
static void on_stop_signal(int sig)
{
    (void)sig;
    stop_requested = 1;
}
static void install_signals(void)
{
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sigemptyset(&sa.sa_mask);
    sa.sa_handler = on_stop_signal;
    sigaction(SIGTERM, &sa, NULL);
    sigaction(SIGQUIT, &sa, NULL);
    sigaction(SIGINT, &sa, NULL);
    sa.sa_handler = SIG_IGN;
    sigaction(SIGPIPE, &sa, NULL);
}
/* pthread_cond_wait that wakes up ten times a second to look at stop_requested (a signal does not wake a condition wait) */
static void cond_wait_tick(pthread_cond_t *cv, pthread_mutex_t *mu)
{
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    ts.tv_nsec += 100 * 1000 * 1000;
    if (ts.tv_nsec >= 1000000000L) {
        ts.tv_sec++;
        ts.tv_nsec -= 1000000000L;
    }
    pthread_cond_timedwait(cv, mu, &ts);
}

/* Packets leave in batches of up to 64 KiB: one fwrite / send per batch. */
static int write_frames(sink *out, const adsb_frame *fr, long n, int outformat)
{
    static char batch[65536 + 256];
    size_t fill = 0;
    unsigned long packets = 0;
    for (long i = 0; i < n; i++) {
        fill += (size_t)adsb_format_frame(&fr[i], outformat, batch + fill);
        packets++;
        if (fill >= 65536 || i + 1 == n) {
            const int rc = sink_write(out, batch, fill, packets);
            if (rc < 0)
                return -1;
            fill = 0, packets = 0; /* (rc == 1: the peer went away and the batch with it, output.c:321-327) */
        }
    }
    return 0;
}

static int flush_frames(adsb_decoder *dec, int outformat)
{
    const adsb_frame *fr; /* the handle's own queue: formatted where it lies */
    const long n = adsb_take(dec, &fr);
    if (n < 0)
        return -1;
    return write_frames(&out_sink, fr, n, outformat);
}

static void print_stats(const adsb_stats *st) /* valid.c:84-100 */
{
    unsigned long long tot = st->ok[0] + st->ok[1] + st->ok[2];
    fprintf(stderr, "\t%10d\t%10d\t%10d\n", 11, 17, 18);
    fprintf(stderr, "Try :\t%10llu\t%10llu\t%10llu\n", (unsigned long long)st->try_[0], (unsigned long long)st->try_[1],
            (unsigned long long)st->try_[2]);
    fprintf(stderr, "Ok :\t%10llu\t%10llu\t%10llu\n", (unsigned long long)st->ok[0], (unsigned long long)st->ok[1],
            (unsigned long long)st->ok[2]);
    fprintf(stderr, "Total :\t%10llu\n", tot); /* tot_fi is uninitialised there */
}

static int write_frames_file(FILE *out, const adsb_frame *fr, long n, int outformat)
{
    char pkt[256];
    for (long i = 0; i < n; i++) {
        int len = adsb_format_frame(&fr[i], outformat, pkt);
        if (fwrite(pkt, 1, (size_t)len, out) != (size_t)len)
            return -1;
    }
    return 0;
}

#define MAX_GPUS 64
#define MAX_FILES 64

/* "-G 4" -> 0,1,2,3; "-G 0,2,2" -> as listed.  Returns the count, 0 on a malformed argument. */
static int parse_gpus(const char *arg, int *devs)
{
    int n = 0;
    if (!strchr(arg, ',')) {
        char *end;
        long k = strtol(arg, &end, 10);
        if (*end || k < 1 || k > MAX_GPUS)
            return 0;
        for (n = 0; n < k; n++)
            devs[n] = n;
        return n;
    }
    for (const char *p = arg; *p;) {
        char *end;
        long k = strtol(p, &end, 10);
        if (end == p || k < 0 || k > 1023 || n == MAX_GPUS)
            return 0;
        devs[n++] = (int)k;
        p = (*end == ',') ? end + 1 : end;
        if (*end && *end != ',')
            return 0;
    }
    return n;
}

/* Narrow the runtime's view to the devices in use, BEFORE its first call (it initialises every device it sees), and
 * renumber devs[] to the ordinals the runtime will then hand out.  A caller's own ROCR_VISIBLE_DEVICES is left alone. */
static void restrict_visible_devices(int *devs, int n)
{
    if (getenv("ROCR_VISIBLE_DEVICES") || getenv("HIP_VISIBLE_DEVICES") || getenv("ADSB_CLI_ALL_DEVICES"))
        return;
    int uniq[MAX_GPUS], nu = 0;
    char list[8 * MAX_GPUS] = "";
    for (int i = 0; i < n; i++) {
        int at = -1;
        for (int k = 0; k < nu; k++)
            if (uniq[k] == devs[i])
                at = k;
        if (at < 0) {
            at = nu;
            uniq[nu++] = devs[i];
            snprintf(list + strlen(list), sizeof list - strlen(list), "%s%d", nu > 1 ? "," : "", devs[i]);
        }
        devs[i] = at;
    }
    setenv("ROCR_VISIBLE_DEVICES", list, 1);
}

/* -G: the library's multi-GPU driver.  One file: sharded over the devices, same bytes as the one-device run.  Several: one
 * capture per device, packets into <file>.<format>. */
static int run_multi(const adsb_config *cfg, int *devs, int ndev, char **files, int nfiles, int outformat, int timing)
{
    const double t_start = now_ms();
    adsb_multi *m = adsb_multi_create(cfg, ndev, devs);
    if (!m) {
        fprintf(stderr, "adsb_multi_create() failed: %s\n", adsb_multi_last_error(NULL));
        return 255;
    }
    const double t_init = now_ms();
    int rc = 0;
    if (nfiles == 1) {
        const adsb_frame *fr = NULL;
        const long n = adsb_multi_decode_file(m, files[0], &fr);
        adsb_stats st;
        if (n < 0) {
            fprintf(stderr, "adsb_multi_decode_file() failed: %s\n", adsb_multi_last_error(m));
            rc = 255;
        } else {
            if (write_frames(&out_sink, fr, n, outformat) != 0)
                rc = 1;
            if (sink_close(&out_sink) != 0)
                rc = 1;
            if (timing) {
                adsb_multi_info inf;
                adsb_multi_get_info(m, &inf);
                fprintf(stderr, "timing: runtime init %.1f ms, decode %.1f ms (%d shards, slowest worker %.1f ms, stitch + gather %.0f us%s), total %.1f ms\n",
                        t_init - t_start, now_ms() - t_init, inf.shards, inf.workers_ms, inf.serial_us,
                        inf.fallback ? ", FELL BACK to one device" : "", now_ms() - t_start);
            }
            if (adsb_multi_get_stats(m, &st) == 0)
                print_stats(&st);
        }
    } else {
        if (adsb_multi_decode_streams_file(m, nfiles, (const char *const *)files) != 0) {
            fprintf(stderr, "adsb_multi_decode_streams_file() failed: %s\n", adsb_multi_last_error(m));
            rc = 255;
        }
        static const char *ext[3] = {"avr", "mlat", "beast"};
        for (int k = 0; k < nfiles && rc == 0; k++) {
            const adsb_frame *fr = NULL;
            const long n = adsb_multi_stream_frames(m, k, &fr);
            char path[4096];
            snprintf(path, sizeof path, "%s.%s", files[k], ext[outformat]);
            FILE *out = n >= 0 ? fopen(path, "wb") : NULL;
            if (!out || write_frames_file(out, fr, n, outformat) != 0 || fclose(out) != 0) {
                fprintf(stderr, "%s: cannot write\n", path);
                rc = 1;
                break;
            }
            adsb_stats st;
            fprintf(stderr, "== %s: %ld frames -> %s\n", files[k], n, path);
            if (adsb_multi_stream_stats(m, k, &st) == 0)
                print_stats(&st);
        }
        if (timing)
            fprintf(stderr, "timing: runtime init %.1f ms, decode %.1f ms, total %.1f ms\n", t_init - t_start, now_ms() - t_init,
                    now_ms() - t_start);
    }
    fflush(stderr);
    if (getenv("ADSB_CLI_CLEAN_EXIT"))
        exit(rc);
    _exit(rc); /* (no teardown: see the end of main) */
}

int main(int argc, char **argv)
{
    const char *filename = NULL;
    char *files[MAX_FILES];
    int nfiles = 0, devs[MAX_GPUS], ndev = 0, device = -1;
    int outformat = 0, df18 = 0, fix1 = 0, c;
    int outmode = SINK_STDOUT;
    const char *rawaddr = NULL;

    while ((c = getopt(argc, argv, "f:g:ambxd:G:s:l:")) != EOF) {
        switch (c) {
        case 'f':
            filename = optarg;
            if (nfiles == MAX_FILES) { /* (more captures than this program keeps track of: say so, do not drop them) */
                usage();
                return 1;
            }
            files[nfiles++] = optarg;
            break;
        case 'd': {
            char *end;
            long k = strtol(optarg, &end, 10);
            if (*end || k < 0 || k > 1023) {
                usage();
                return 1;
            }
            device = (int)k;
            break;
        }
        case 'G':
            ndev = parse_gpus(optarg, devs);
            if (ndev == 0) {
                usage();
                return 1;
            }
            break;
        case 's':
            rawaddr = optarg;
            outmode = SINK_CONNECT;
            break;
        case 'l':
            rawaddr = optarg;
            outmode = SINK_LISTEN;
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
    /* several captures need -G and go to files; -d and -G exclude each other */
    if (!filename || (nfiles > 1 && (ndev == 0 || outmode != SINK_STDOUT)) || (ndev && device >= 0)) {
        usage();
        return 1;
    }
    install_signals();
    sink_init(&out_sink, outmode, rawaddr);
    out_sink.stop = &stop_requested;
    if (getenv("ADSB_CLI_RETRY_S")) /* (a test's knob: the reference waits 3 s between attempts, output.c:282) */
        out_sink.retry_s = (unsigned)atoi(getenv("ADSB_CLI_RETRY_S"));

    const int timing = getenv("ADSB_CLI_TIMING") ? (atoi(getenv("ADSB_CLI_TIMING")) > 1 ? 2 : 1) : 0;
    if (ndev) {
        adsb_config mcfg;
        adsb_config_default(&mcfg);
        mcfg.df18 = df18;
        mcfg.fix_1bit = fix1;
        mcfg.collect_stats = 1; /* the reference always prints Try/Ok */
        /* The driver cuts FILES into slices (every device preads its own): what the reference accepts beyond that goes the
         * way the one-device run goes it -- a single -f that is a pipe, a FIFO or a device node is streamed through one
         * device (the first one listed), a single -f that cannot be opened ends the run silently with an empty table
         * (air.c:225-228), same bytes on stdout and stderr either way. */
        struct stat sb;
        const int one_ok = nfiles > 1 || (stat(files[0], &sb) == 0 && S_ISREG(sb.st_mode) && access(files[0], R_OK) == 0);
        if (one_ok) {
            restrict_visible_devices(devs, ndev);
            const int prc = sink_wait_peer(&out_sink);
            if (prc == 2) { /* told to end before a peer came: nothing decoded, the table (all zero) and status 0 (main.c:101-105) */
                adsb_stats z;
                memset(&z, 0, sizeof z);
                print_stats(&z);
                return 0;
            }
            if (prc != 0)
                return 255; /* unusable address: runOutput() == -1 (output.c:278-279) */
            return run_multi(&mcfg, devs, ndev, files, nfiles, outformat, timing);
        }
        device = devs[0];
        ndev = 0;
    }
    if (device >= 0) {
        restrict_visible_devices(&device, 1);
    } else if (!getenv("ROCR_VISIBLE_DEVICES") && !getenv("HIP_VISIBLE_DEVICES") && !getenv("ADSB_CLI_ALL_DEVICES")) {
        int first = 0;
        restrict_visible_devices(&first, 1); /* one device is all this run uses: do not start the others */
    }
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
        unsigned long long ring_max = RING_MAX_BYTES;
        if (getenv("ADSB_CLI_RING_MB") && atoll(getenv("ADSB_CLI_RING_MB")) >= 96) /* (measurement knob of this program, not of the library) */
            ring_max = (unsigned long long)atoll(getenv("ADSB_CLI_RING_MB")) << 20;
        if (size > ring_max || size == 0)
            size = ring_max; /* pipes and large files: a bounded ring, the reader waits for free buffers */
        rg.nbuf = (int)(size / ((unsigned long long)BUF_SAMPLES * 2)) + 2;
        if (rg.nbuf < 3)
            rg.nbuf = 3;
        rg.slot = (ring_slot *)calloc((size_t)rg.nbuf, sizeof *rg.slot);
        pthread_mutex_init(&rg.mu, NULL);
        pthread_cond_init(&rg.cv, NULL);
        have_reader = rg.slot && pthread_create(&reader, NULL, reader_main, &rg) == 0;
    }

    /* -s / -l: the peer first (output.c:277-285: no peer, no packets; an unusable address ends the run with
     * runOutput() == -1).  The file is being read meanwhile. */
    {
        const int prc = sink_wait_peer(&out_sink);
        if (prc == 2) { /* told to end before a peer came */
            adsb_stats z;
            memset(&z, 0, sizeof z);
            print_stats(&z);
            fflush(stderr);
            _exit(0);
        }
        if (prc != 0)
            return 255;
    }

    adsb_config cfg;
    adsb_config_default(&cfg);
    cfg.df18 = df18;
    cfg.fix_1bit = fix1;
    cfg.collect_stats = 1; /* the reference always prints Try/Ok */
    cfg.device = device;   /* (-1: the current one; after the narrowing above, 0) */
    cfg.warm_start = 1;    /* a one-shot process: the runtime's first-use costs are paid inside adsb_create, beside its other work */
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
    int stopped = 0, n_push = 0;
    double t_wait_ring = 0, t_push = 0, t_flush = 0, t_finish = 0;
    if (have_reader) {
        int prev = -1;
        for (int k = 0;; k++) {
            ring_slot *s = &rg.slot[k % rg.nbuf];
            const double t_w0 = now_ms();
            pthread_mutex_lock(&rg.mu);
            while (!s->ready && !stop_requested)
                cond_wait_tick(&rg.cv, &rg.mu);
            const int reader_failed = rg.failed; /* (written under the mutex by the reader) */
            const int is_ready = s->ready;
            pthread_mutex_unlock(&rg.mu);
            t_wait_ring += now_ms() - t_w0;
            if (stop_requested && !is_ready) {
                stopped = 1; /* SIGINT / SIGTERM / SIGQUIT: no more pushes; what is in the decoder is finished below */
                break;
            }
            if (reader_failed) {
                fprintf(stderr, "out of memory for the read buffers\n");
                rc = 255;
                break;
            }
            const size_t bytes = s->bytes;
            if (bytes >= 2) {
                /* a trailing odd byte is dropped, like decodeiq(iqbuff, n / 2) (air.c:239) */
                const double t_p0 = now_ms();
                const int prc = s->registered ? adsb_push_async(dec, s->buf, bytes / 2) : adsb_push(dec, s->buf, bytes / 2);
                t_push += now_ms() - t_p0;
                n_push++;
                if (timing > 1)
                    fprintf(stderr, "  push %d (%s): %.2f ms at +%.1f ms\n", n_push, s->registered ? "async" : "sync", now_ms() - t_p0, t_p0 - t_init);
                if (prc != 0) {
                    fprintf(stderr, "adsb_push() failed: %s\n", adsb_last_error(dec));
                    rc = 255;
                    break;
                }
                const double t_f0 = now_ms();
                const int frc = flush_frames(dec, outformat);
                t_flush += now_ms() - t_f0;
                if (frc != 0) { /* stdout is gone (EPIPE, disk full): stop, and say so */
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
            if (stop_requested) {
                stopped = 1;
                break;
            }
        }
        const double t_fin0 = now_ms();
        if (rc == 0 && adsb_finish(dec) != 0) {
            fprintf(stderr, "adsb_finish() failed: %s\n", adsb_last_error(dec));
            rc = 255;
        }
        if (flush_frames(dec, outformat) != 0 && rc == 0)
            rc = 1;
        t_finish = now_ms() - t_fin0;
        if (sink_close(&out_sink) != 0 && rc == 0)
            rc = 1; /* the last block did not reach the file (ENOSPC, EIO, a closed pipe) */
        if (rc != 0 || stopped) { /* let the reader run out: hand every buffer back */
            pthread_mutex_lock(&rg.mu);
            for (int i = 0; i < rg.nbuf; i++)
                rg.slot[i].ready = rg.slot[i].filled = 0;
            pthread_cond_broadcast(&rg.cv);
            pthread_mutex_unlock(&rg.mu);
            close(rg.fd); /* read() fails from here on */
        }
        if (rc == 0 && !stopped) { /* (after a failure or a signal the threads are left to the process exit) */
            pthread_join(reader, NULL);
            pthread_join(locker, NULL);
            close(rg.fd);
        }
    }
    const double t_done = now_ms();
    if (timing)
        fprintf(stderr, "timing: runtime init %.1f ms, decode %.1f ms (page-locking, on its own thread: %.1f ms; main thread: %d pushes %.1f ms, "
                        "waiting for a read + locked buffer %.1f ms, formatting + writing %.1f ms, finish %.1f ms), total %.1f ms\n",
                t_init - t_start, t_done - t_init, rg.t_reg, n_push, t_push, t_wait_ring, t_flush, t_finish, t_done - t_start);

    adsb_stats st;
    if (adsb_get_stats(dec, &st) == 0)
        print_stats(&st);
    /* no adsb_destroy / unregister / free: the process ends here, and tearing the GPU runtime
     * down cleanly costs tens of milliseconds that an offline decode has no use for */
    fflush(stderr);
    if (getenv("ADSB_CLI_CLEAN_EXIT")) /* (a profiler's knob: rocprofv3 writes its trace from an exit handler) */
        exit(rc);
    _exit(rc);
}
