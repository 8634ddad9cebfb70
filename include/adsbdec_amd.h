/*
 * adsbdec_amd.h -- C-ABI of libadsbdec_amd.so: the MI355X (gfx950) drop-in for the offline "-f" demodulation path
 * of TLeconte/adsbdec.  This header is everything a drop-in host and a multi-GPU host call; the primitives underneath
 * (resolver handle, hand-off walker, shard scans, stitcher) and the test knobs are in adsbdec_amd_diag.h.
 *
 * The reference has no plugin/FFI interface: its whole interface is one prototype (adsbdec.h:5) plus extern C functions
 * with file-scope state.  Each entry point names the reference seam it stands behind (file:line under
 * /root/reference); INTEGRATION.md shows the change a maintainer makes in air.c / output.c to bind them.
 *
 * Conventions follow the reference: int 0 / -1 with a message from adsb_last_error() (the reference prints to stderr,
 * air.c:113-118); one producer thread per handle (decodeiq is not re-entrant: air.c:33-34,49-50, demod.c:86); plain
 * pointers and sizes, no C++/torch types.  The HIP path is the only implementation: there is no CPU fallback, and
 * adsb_create() fails loudly when no gfx950 device is usable.
 *
 * Input domain.  uint16 samples carrying the Airspy's 12-bit ADC code centred on 2048 (air.c:64).  Results are
 * bit-identical to the reference for every code in [0, 4095] and beyond, up to |x-2048| <= ~23 000 (the preamble sums
 * still fit an int); larger codes make the reference's `int p1 = float + float` (demod.c:102-105) overflow -- undefined
 * behaviour that wraps with gcc -- while this library compares un-wrapped values: accepted, no parity claimed.
 * Streams are limited to < 2^32 samples (the reference's `fidx` wraps there): a push that would reach it fails.
 */
#ifndef ADSBDEC_AMD_H
#define ADSBDEC_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped when an entry point or struct member changes meaning or place; within one version adsb_config and adsb_profile grow
 * at their END only (adsb_create reads cfg->struct_size bytes, adsb_get_profile passes the caller's size).
 * 5 (round 6): adsb_config carries `abi` (adsb_create / adsb_multi_create refuse any other value by name: a binary built against
 *    ABI <= 4 must be rebuilt); the debug_* test knobs left it for adsb_debug_config (adsbdec_amd_diag.h); adsb_profile grew. */
#define ADSB_ABI_VERSION 5

/* Constants of the path (adsbdec.h:1-3, air.c:32,47). */
#define ADSB_PULSEW 5
#define ADSB_DECOFFSET 1200
#define ADSB_APBUFFSZ 40980
#define ADSB_WINDOW 1196 /* power samples one long-frame evaluation touches: a[g .. g+1195] */

typedef struct adsb_decoder adsb_decoder; /* one stream == the statics of air.c / demod.c / valid.c */

/* Record leaving the path == the arguments of netout() (valid.c:26, output.c:159) == blk_t (output.c:45-52), plus the
 * global power-sample index of the preamble. */
typedef struct adsb_frame {
    uint64_t g;        /* global 10 MS/s power-sample index of the preamble start */
    uint64_t ts;       /* demod.c:86,99: loop-pass counter at acceptance          */
    uint32_t pw;       /* demod.c:127,133: (p1+p2)/4                              */
    uint8_t len;       /* 7 (DF11) or 14 (DF17/18)                                */
    uint8_t frame[14]; /* demod.c:110-123                                         */
    uint8_t reserved;  /* bit 0: repaired by the 1-bit extension (cfg.fix_1bit)   */
} adsb_frame;

/* valid.c:30-31,84-100: Try/Ok per DF, in the order 11, 17, 18. */
typedef struct adsb_stats {
    uint64_t try_[3];
    uint64_t ok[3];
    uint64_t fixed; /* frames accepted after a 1-bit repair (extension; 0 by default) */
} adsb_stats;

typedef struct adsb_config {
    uint32_t struct_size;   /* sizeof(adsb_config) as the CALLER knows it                                         */
    uint32_t abi;           /* ADSB_ABI_VERSION of the caller's header; anything else is refused                  */
    int32_t df18;           /* demod.c:26 `df`, set by -a (main.c:76-78)                                          */
    int32_t device;         /* HIP device ordinal; -1 = the current device                                        */
    int32_t collect_stats;  /* reproduce valid.c's Try counters (the reference always does; costs a try list)     */
    int32_t profile;        /* time every scan launch on the device's own 100 MHz clock: adsb_profile.kernel_ms   */
    uint64_t stage_samples; /* device staging capacity for adsb_push(); 0 = default (32 Mi)                       */
    void *stream;           /* hipStream_t to launch on; NULL = a stream owned by the handle                      */
    int32_t all_candidates; /* 1: the device reports EVERY CRC-valid offset; 0 (default): it drops those the greedy
                               scan can provably never visit (same frames, ~4x fewer records)                     */
    int32_t fix_1bit;       /* EXTENSION, not in the reference (its -e does nothing): repair DF17/18 frames
                               whose CRC residual is the syndrome of one bit in [5,112).  Off by default.         */
    int32_t push_overlap;   /* 1: adsb_push() returns once `samples` is COPIED to the device and leaves the scan in
                               flight; the frames of a call become drainable during the NEXT push / finish / sync
                               (same frames, same order).  0 (default): drainable when the call returns.          */
    int32_t host_threads;   /* Threads that consume the device's hand-off stream (decodeiq, air.c:54, never started one).
                               1: the calling thread alone, ALWAYS -- the library never starts a thread.
                               2: + a thread of the handle's own that reads the stream of large launches; N >= 3
                               (<= 17): + N - 2 more that decide batches of tiles ahead and write the frames.
                               0 (default): 1, until a launch hands over a record per 2 048 offsets (a channel near its capacity);
                               launches that follow such a one run with 6 (5 threads that poll during a launch and
                               0.4 ms beyond, then sleep; 2 where the process has < 12 CPUs): the step takes 1.2 x its
                               kernel instead of 3.3 x.  adsb_profile.host_threads_running says what exists; same
                               frames, order and counters whatever the value.                     */
    int32_t wait_timeout_s; /* no wait for the device lasts longer (0 = default, 120 s): a launch or copy that never
                               completes ends the call with -1 and adsb_last_error() names what was waited for     */
    int32_t warm_start;     /* 1: adsb_create also pays the runtime's first-use costs of copying (first large copy,
                               second copy engine: 7-9 ms each) beside its other work: for a one-shot process      */
    const void *debug;      /* NULL, or an adsb_debug_config (adsbdec_amd_diag.h: test knobs); copied by adsb_create */
} adsb_config;

/* Counters accumulate over the life of the handle (adsb_reset keeps them: take differences). */
typedef struct adsb_profile {
    uint64_t launches;     /* scan-kernel launches since adsb_create                                  */
    uint64_t relaunches;   /* launches repeated after a record-buffer overflow                        */
    uint64_t offsets;      /* preamble offsets those launches covered                                 */
    double kernel_ms;      /* sum of their durations on the device clock (profile=1 only)             */
    double last_kernel_ms;
    uint64_t last_offsets;
    uint64_t candidates;   /* CRC-valid candidates received from the device                           */
    uint64_t tries;        /* DF-gate passes that came through launch-wide lists (collect_stats=1)    */
    double host_ms;        /* host time spent checking + resolving records                            */
    double wait_ms;        /* host time blocked waiting for the device                                */
    uint64_t big_offsets;  /* offsets per launch of the largest launch size seen                      */
    uint64_t big_launches; /* launches of that size                                                   */
    double big_ms;         /* sum of their durations on the device clock (profile=1)                  */
    /* ---- ABI 5 ---- */
    uint32_t host_threads_running; /* threads of the handle's own that exist NOW: reader (0/1) + gang helpers; 0 under
                                      ordinary traffic and always with cfg.host_threads = 1                        */
    uint32_t gang_launches;        /* launches whose frames went through the gang                                  */
    uint64_t gang_batches;         /* batches of tiles handed to the gang to be decided ahead of the caller        */
} adsb_profile;

/* Defaults; the struct's size is the CALLER's: adsb_config_default(&cfg) = adsb_config_init(&cfg, sizeof cfg), which sets cfg.abi. */
void adsb_config_init(adsb_config *cfg, size_t struct_size);
#define adsb_config_default(cfg) adsb_config_init((cfg), sizeof(adsb_config))

/* The stream state that air.c:33-34,49-50 / demod.c:86 / valid.c:30-31 keep in statics.  NULL on failure
 * (adsb_last_error(NULL) has the reason).  adsb_reset: the same handle, a fresh stream (ring, ts, stats). */
adsb_decoder *adsb_create(const adsb_config *cfg);
void adsb_destroy(adsb_decoder *d);
int adsb_reset(adsb_decoder *d);
/* Sample ingress; replaces `decodeiq(const unsigned short *r, const int len)` (air.c:54), called from fileInput
 * (air.c:239) / rx_callback (air.c:175).  `samples` is borrowed for the call.  Any n is accepted; the stream is the
 * concatenation of all pushes (the reference requires n % 4 == 0). */
int adsb_push(adsb_decoder *d, const uint16_t *samples, size_t n);
/* The same from a double-buffered read loop (fileInput with two iqbuffs): returns once copy and scan of this chunk are
 * ENQUEUED, then collects the PREVIOUS chunk's frames.  `samples` stays borrowed until the next push / finish / sync on
 * the handle returns; frames become drainable one call later, never reordered.  adsb_sync waits for everything. */
int adsb_push_async(adsb_decoder *d, const uint16_t *samples, size_t n);
int adsb_sync(adsb_decoder *d);
/* Samples already resident in HBM (no reference counterpart).  A 16-byte aligned pointer at a stream position that is a
 * multiple of 8 samples is scanned in place.  _final = push of the LAST piece + adsb_finish in one pass. */
int adsb_push_device(adsb_decoder *d, const void *device_samples, size_t n);
int adsb_push_device_final(adsb_decoder *d, const void *device_samples, size_t n);
/* The whole of `adsbdec -f` for ONE capture resident in HBM: adsb_reset + adsb_push_device_final + adsb_take.
 * Returns the number of frames (*frames as adsb_take), or -1. */
long adsb_decode_device(adsb_decoder *d, const void *device_samples, size_t n, const adsb_frame **frames);
/* End of input (fileInput's EOF, air.c:241-244): the remaining offsets, and the end-of-file horizon. */
int adsb_finish(adsb_decoder *d);
/* Page-locked host buffers: the counterpart of `iqbuff = malloc(...)` (air.c:230), so that a push is one DMA.
 * adsb_host_alloc_on binds the memory to the NUMA node of `device` (two-socket hosts; best effort).  adsb_host_register
 * page-locks memory the caller already owns.  0 / -1. */
void *adsb_host_alloc(size_t bytes);
void *adsb_host_alloc_on(size_t bytes, int device);
void adsb_host_free(void *p);
int adsb_host_register(void *p, size_t bytes);
int adsb_host_unregister(void *p);
/* Frame egress: the records the reference hands to netout() (output.c:159), in the same order.  adsb_drain copies
 * (returns the number, <= cap, or -1); adsb_take hands every pending frame out in place -- valid until the next call
 * that pushes into, finishes, resets or destroys the handle. */
long adsb_drain(adsb_decoder *d, adsb_frame *out, size_t cap);
long adsb_take(adsb_decoder *d, const adsb_frame **frames);
size_t adsb_pending(const adsb_decoder *d);
/* print_stats() counters (valid.c:84-100); try_ needs collect_stats=1 (counted on the device, fetched by this call). */
int adsb_get_stats(const adsb_decoder *d, adsb_stats *out);
/* Fills the first `size` bytes of *out (the macro passes the caller's sizeof: adsb_profile grows at its end). */
int adsb_get_profile_sized(const adsb_decoder *d, adsb_profile *out, size_t size);
#define adsb_get_profile(d, out) adsb_get_profile_sized((d), (out), sizeof(adsb_profile))
/* Last error text of a handle, or of the last failed adsb_create() when d == NULL. */
const char *adsb_last_error(const adsb_decoder *d);
/* formatpkt() (output.c:204-262, WITH_AIR).  outformat 0 = AVR "*hex;\n", 1 = AVR-MLAT "@ts48hex;\n", 2 = Beast.
 * pkt must hold 256 bytes.  Returns the packet length. */
int adsb_format_frame(const adsb_frame *f, int outformat, char *pkt);
/* The CPUs local to HIP device `device` (/sys/bus/pci/devices/<bdf>/local_cpulist, e.g. "0-63,128-191") and its NUMA
 * node: the thread that feeds a handle polls memory the device writes; on the far socket it was measured 2.5-3 x slower.
 * Length of the string, 0 when the platform does not say, -1 on error / node or -1. */
int adsb_device_cpulist(int device, char *out, size_t cap);
int adsb_device_numa_node(int device);
/* Shard planning: splits the offsets [0, power_samples - ADSB_WINDOW] of one stream over n_shards owners.
 * Shard i owns offsets [g_begin[i], g_end[i]) (g_begin a multiple of 28) and must be given the input samples
 * [first_sample[i], first_sample[i] + n_samples[i]) -- 2 408 samples of halo.  Returns the shards used (<= n_shards). */
int adsb_plan_shards(uint64_t total_samples, int n_shards, uint64_t *g_begin, uint64_t *g_end,
                     uint64_t *first_sample, uint64_t *n_samples);
/* ---- ONE process, several GPUs (csrc/multi.cpp): the host of BASELINE configs[3] / configs[4] ---------------------
 * A worker thread and a decoder handle per device; no collective on the data path.  Stands where
 * fileInput's loop (air.c:217-246) hands its buffers to decodeiq and the frames come back in ascending order for
 * netout (output.c:159-182). */
typedef struct adsb_multi adsb_multi;
typedef struct adsb_multi_info { /* of the last adsb_multi_decode_* call */
    int32_t shards;         /* shards the capture was cut into (streams decoded side by side, for the stream calls)  */
    int32_t fallback;       /* 1: a seam could not be decided from the head candidates; the capture then went through
                               ONE device as an ordinary stream (same frames)                                       */
    uint64_t calls_walked;  /* deqframe calls the end-of-file walk replayed on the calling thread ...                */
    uint64_t calls_jumped;  /* ... and skipped by jumping onto the shards' own walks                                 */
    double create_ms;       /* adsb_multi_create: the slowest worker's adsb_create (the devices start side by side)  */
    double workers_ms;      /* the slowest worker's share of the call                                                */
    double stitch_us;       /* the stitcher on the calling thread                                                    */
    double serial_us;       /* everything behind the last worker: stitch + gather into one array                     */
    double total_ms;
    int32_t workers_bound;  /* workers whose thread runs on the CPUs of its device's NUMA node                       */
    int32_t helper_threads; /* reader / gang threads the workers' handles own after the call (sum of
                               adsb_profile.host_threads_running; per worker: adsb_multi_worker_profile)             */
} adsb_multi_info;

/* n_devices workers; devices[i] = HIP ordinal of worker i (NULL: 0 .. n_devices-1; an ordinal may repeat: plumbing tests
 * on a one-GPU box).  cfg as for adsb_create (device and stream are ignored).  NULL on failure (adsb_multi_last_error(NULL)). */
adsb_multi *adsb_multi_create(const adsb_config *cfg, int n_devices, const int *devices);
void adsb_multi_destroy(adsb_multi *m);
int adsb_multi_devices(const adsb_multi *m);

/* configs[4]: ONE capture, time-sharded over as many devices as it is worth (>= 128 Ki offsets per shard); each worker
 * feeds its halo'd slice in 32 MiB pieces (copy of a piece under the scan of the one before), resolves its shard while
 * its kernels run; the calling thread stitches and the workers gather.  Returns the number of frames, in the reference's
 * order, *frames valid until the next call on m; -1 on failure (after a worker was given up -- adsb_multi_last_error
 * says so -- the handle only answers -1, and the SOURCE buffers of that call must stay alive: the worker may come back).
 *   _host:   the capture lies in host memory (page-lock it, or every piece goes through the runtime's bounce buffers);
 *   _file:   every worker reads its own slice of a regular file into page-locked buffers of its own;
 *   _device: slice i is resident in the HBM of worker i's device and holds the samples adsb_multi_plan says. */
long adsb_multi_decode_host(adsb_multi *m, const uint16_t *samples, size_t n, const adsb_frame **frames);
long adsb_multi_decode_file(adsb_multi *m, const char *path, const adsb_frame **frames);
long adsb_multi_decode_device(adsb_multi *m, uint64_t total_samples, const void *const *slices, int n_slices,
                              const adsb_frame **frames);
int adsb_multi_plan(const adsb_multi *m, uint64_t total_samples, uint64_t *g_begin, uint64_t *g_end,
                    uint64_t *first_sample, uint64_t *n_samples);
int adsb_multi_get_stats(const adsb_multi *m, adsb_stats *out); /* the stream's Try/Ok table (cfg.collect_stats) */

/* configs[3]: n_streams INDEPENDENT captures, stream s on worker s mod adsb_multi_devices(m), each with its own ts and
 * statistics -- N times what `adsbdec -f` does (main.c:60-89), side by side.  0 / -1; results per stream afterwards. */
int adsb_multi_decode_streams_host(adsb_multi *m, int n_streams, const uint16_t *const *samples, const size_t *n);
int adsb_multi_decode_streams_file(adsb_multi *m, int n_streams, const char *const *paths);
long adsb_multi_stream_frames(const adsb_multi *m, int stream, const adsb_frame **frames);
int adsb_multi_stream_stats(const adsb_multi *m, int stream, adsb_stats *out);

int adsb_multi_get_info(const adsb_multi *m, adsb_multi_info *out);
/* A page-locked array for ONE capture that adsb_multi_decode_host will decode, every shard's part on the NUMA node of the
 * device that pulls it (free with adsb_host_free); adsb_multi_worker_placement says whether that held for the last decode. */
uint16_t *adsb_multi_host_alloc(adsb_multi *m, uint64_t total_samples);
typedef struct adsb_worker_placement {
    int32_t device, device_node; /* the worker's HIP device and its NUMA node (-1: the platform does not say)      */
    int32_t thread_bound;        /* the worker's thread runs on the CPUs of that node                              */
    int32_t slice_node;          /* the node most pages of the worker's slice of the last host capture live on     */
    double local_fraction;       /* share of the sampled pages of that slice on device_node (1.0 = fed from its socket) */
} adsb_worker_placement;
int adsb_multi_worker_placement(const adsb_multi *m, int worker, adsb_worker_placement *out);
/* adsb_get_profile of worker `worker`'s handle (first `size` bytes; the macro passes the caller's sizeof). */
int adsb_multi_worker_profile_sized(const adsb_multi *m, int worker, adsb_profile *out, size_t size);
#define adsb_multi_worker_profile(m, worker, out) adsb_multi_worker_profile_sized((m), (worker), (out), sizeof(adsb_profile))
/* Last error text of m, or of the last failed adsb_multi_create() when m == NULL; names the device and worker. */
const char *adsb_multi_last_error(const adsb_multi *m);

int adsb_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
