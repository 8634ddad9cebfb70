/*
 * adsbdec_amd.h -- C-ABI of libadsbdec_amd.so: the MI355X (gfx950) drop-in for the
 * offline "-f" demodulation path of TLeconte/adsbdec.
 *
 * The reference has no plugin/FFI interface; its seams are plain extern C
 * functions with file-scope state.  Each entry point below names
 * the reference interface it stands behind (file:line under /root/reference).
 * INTEGRATION.md shows the three-line change a maintainer makes in air.c /
 * output.c to bind them.
 *
 * Conventions follow the reference: int 0 / -1 with a message retrievable through
 * adsb_last_error() (the reference prints to stderr, air.c:113-118); one producer
 * thread per handle (decodeiq is not re-entrant, air.c:33-34,49-50; demod.c:86).
 * Plain pointers and sizes only; no C++/torch types cross this boundary.
 *
 * The HIP path is the only implementation behind these symbols: there is no CPU
 * fallback, and adsb_create() fails loudly when no gfx950 device is usable.
 *
 * Input domain.  Samples are uint16 carrying the Airspy's 12-bit ADC code centred on
 * 2048 (air.c:64 `(float)r[i]-0x800`).  Results are bit-identical to the reference for
 * every code in [0, 4095] and, beyond the ADC's range, for codes up to ~25 000
 * (tested to 32 000 with |x-2048| <= ~23 000): there the preamble sums still fit an
 * int.  Larger codes make the reference's `int p1 = float + float` (demod.c:102-105)
 * and `2*s1` overflow -- undefined behaviour in C that happens to wrap with gcc --
 * while this library compares the un-wrapped values; adsb_push accepts such samples
 * but no parity is claimed for them.  Streams are limited to < 2^32
 * samples (the reference's `fidx` wraps there): a push that would reach 2^32 fails.
 */
#ifndef ADSBDEC_AMD_H
#define ADSBDEC_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped when an existing entry point or struct member changes meaning.  Additions do not bump it: new entry points are
 * new symbols, and adsb_config grows at its end only -- adsb_create() reads no further than cfg->struct_size, so a caller
 * built against a shorter adsb_config keeps working (the members it does not know default to 0).
 * 3: `push_overlap` took the place of ABI 2's `reserved0`; it is only honoured when struct_size covers `host_threads`
 *    (a caller built against ABI 2 that left garbage in reserved0 keeps ABI 2's behaviour).
 * 4: adsb_shard_head / adsb_shard_part carry the statistics of a resolved shard (round 4): adsb_shard_head grew from 80 to
 *    136 bytes and adsb_shard_part by four members, and the library fills / reads all of them -- a binary built against
 *    ABI 3 that calls the shard API (adsb_scan_shard_resolved*, adsb_shard_end, adsb_stitch_shards*) MUST be rebuilt; it
 *    can find out at run time: adsb_shard_layout_check(sizeof(adsb_shard_head), sizeof(adsb_shard_part)) != 0.  The
 *    streaming API (adsb_create / adsb_push* / adsb_drain ...) and adsb_multi_* are unaffected.
 *    adsb_stitch_shards* answer -2 (not -1) when new_cap is too small (round 5). */
#define ADSB_ABI_VERSION 4

/* Constants of the path (adsbdec.h:1-3, air.c:32,47). */
#define ADSB_PULSEW 5
#define ADSB_DECOFFSET 1200
#define ADSB_APBUFFSZ 40980
#define ADSB_WINDOW 1196 /* power samples one long-frame evaluation touches: a[g .. g+1195] */

typedef struct adsb_decoder adsb_decoder; /* one stream == the statics of air.c/demod.c/valid.c */

/* Record leaving the path == the arguments of netout() (valid.c:26, output.c:159)
 * == blk_t (output.c:45-52), plus the global power-sample index of the preamble. */
typedef struct adsb_frame {
    uint64_t g;        /* global 10 MS/s power-sample index of the preamble start */
    uint64_t ts;       /* demod.c:86,99: loop-iteration counter at acceptance      */
    uint32_t pw;       /* demod.c:127,133: (p1+p2)/4                               */
    uint8_t len;       /* 7 (DF11) or 14 (DF17/18)                                 */
    uint8_t frame[14]; /* demod.c:110-123                                          */
    uint8_t reserved;  /* bit 0: frame was repaired by the 1-bit extension (cfg.fix_1bit) */
} adsb_frame;

/* A CRC-valid candidate before greedy resolution (what one GPU emits for the
 * offsets it owns; the unit the host gathers across shards). */
typedef struct adsb_candidate {
    uint64_t g;
    uint32_t pw;
    uint8_t len;
    uint8_t frame[14];
    uint8_t reserved;
} adsb_candidate;

/* valid.c:30-31,84-100: Try/Ok per DF, in the order 11, 17, 18. */
typedef struct adsb_stats {
    uint64_t try_[3];
    uint64_t ok[3];
    uint64_t fixed;   /* frames accepted after a 1-bit repair (extension; always 0 by default) */
} adsb_stats;

typedef struct adsb_config {
    uint32_t struct_size;      /* sizeof(adsb_config), for forward compatibility */
    int32_t df18;              /* demod.c:26 `df`, set by -a (main.c:76-78)        */
    int32_t device;            /* HIP device ordinal; -1 = the current device     */
    int32_t collect_stats;     /* reproduce valid.c's Try counters (costs a try list) */
    int32_t profile;           /* time every scan launch on the device's own 100 MHz clock (latest tile
                                  end - earliest tile start, taken inside the kernel): adsb_profile */
    int32_t debug_queue_cap;   /* test knob: survivor-queue entries per workgroup round (256..1024); 0 = default */
    uint64_t stage_samples;    /* device staging capacity for adsb_push(); 0 = default (32 Mi) */
    void *stream;              /* hipStream_t to launch on; NULL = a stream owned by the handle */
    int32_t all_candidates;    /* 1: the device reports EVERY CRC-valid offset; 0 (default): it drops the
                                  ones the greedy scan can provably never visit (same frames, ~4x fewer records) */
    int32_t fix_1bit;          /* EXTENSION, not in the reference (its -e flag does nothing):
                                  repair DF17/18 frames whose CRC residual is the syndrome of one bit
                                  in [5,112). Off by default; no reference parity exists for it. */
    /* test knobs (0 = default): shrink the launch-wide record buffers / the per-tile staged list so
     * that tests can force every overflow path (relaunch with regrown buffers, loose list) */
    int32_t debug_cand_cap;    /* loose-list records per launch slot                         */
    int32_t debug_try_cap;     /* try words per launch slot (collect_stats)                  */
    int32_t debug_clist_cap;   /* CRC-valid candidates staged per tile (1..256)              */
    int32_t push_overlap;      /* 1: adsb_push() returns as soon as `samples` has been COPIED to the device (the buffer
                                  is free again, which is all decodeiq's callers need: air.c:230-239, 173-177) and
                                  leaves the scan in flight; the frames of a call become drainable during the NEXT
                                  adsb_push / adsb_finish / adsb_sync instead of during the call itself (same frames,
                                  same order).  0 (default): frames are drainable when the call returns. */
    int32_t host_threads;      /* 1: the calling thread alone consumes the device's hand-off stream.
                                  2: the handle owns a second thread that reads and checks the stream of large launches
                                  while the caller resolves behind it; same frames, same order.  The thread spins for
                                  ~0.4 ms after a launch, then sleeps until the next one.
                                  N >= 3 (at most 17): that thread, and N - 2 more that decide batches of tiles ahead of
                                  the caller (which only takes their decisions over) and write the frames; same frames,
                                  same order, same counters.  They poll during a launch and for ~0.4 ms after it.
                                  0 (default): 1, until a launch hands over 65 536 records or more (a channel near its
                                  capacity: ~20 k frames per second of signal); from the next launch on, 6 for every
                                  launch that follows such a one (BASELINE configs[2]: a 256 Mi-sample step takes
                                  1.3 x its kernel instead of 3.3 x) -- 2 where the process may use fewer than 12 CPUs. */
    /* more test knobs (0 = default).  The library reads no environment variable: whatever a test has to force is here. */
    int32_t debug_no_streaming;     /* 1: every launch is collected after completion (no hand-off stream)              */
    int32_t debug_frames_cap;       /* collect_stats: accepted frames the upload buffers start with (regrow path)      */
    int32_t debug_reader_min_tiles; /* host_threads = 2: launches of at least this many tiles go through the thread    */
    int32_t debug_shard_head;       /* resolved shards: offsets whose candidates are all kept for the stitcher (16384) */
    int32_t debug_passes;           /* passes per tile of every launch (2..32) instead of the cost model's choice      */
    int32_t debug_stagger;          /* leading tiles of staggered size (scan_kernel.h tile_passes)                      */
    int32_t wait_timeout_s;    /* No wait for the device inside the library lasts longer than this many seconds (0 = default,
                                  120): a launch or copy that never completes -- a wedged queue, a lost device -- ends the
                                  call with -1 and adsb_last_error() names what was waited for; the multi-GPU driver gives
                                  its workers the same limit (plus 30 s) and names the worker.  Nothing is retried. */
    int32_t debug_gang_min;    /* test knob: host_threads >= 3 -- batches of at least this many records go through the
                                  threads (0 = default: 2048 records decided ahead, 1024 frames written) */
} adsb_config;

/* Counters accumulate over the life of the handle (adsb_reset keeps them: a caller that
 * decodes many captures takes differences). */
typedef struct adsb_profile {
    uint64_t launches;         /* scan-kernel launches since adsb_create            */
    uint64_t relaunches;       /* launches repeated after a record-buffer overflow  */
    uint64_t offsets;          /* preamble offsets those launches covered           */
    double kernel_ms;          /* sum of their durations on the device clock (profile=1 only) */
    double last_kernel_ms;
    uint64_t last_offsets;
    uint64_t candidates;       /* CRC-valid candidates received from the device     */
    uint64_t tries;            /* DF-gate passes that came through launch-wide lists (collect_stats=1: per-shard scans and
                                * queue-overflow rounds; tries counted from the tiles' own regions on the device are not in it) */
    double host_ms;            /* host time spent sorting + resolving records       */
    double wait_ms;            /* host time blocked waiting for the device          */
    uint64_t big_offsets;      /* offsets per launch of the largest launch size seen */
    uint64_t big_launches;     /* launches of that size                             */
    double big_ms;             /* sum of their durations on the device clock (profile=1) */
} adsb_profile;

/* Defaults.  The struct's size is the CALLER's (the library's own adsb_config may be longer: it grows at its end), so it
 * is passed along: adsb_config_default(&cfg) compiles to adsb_config_init(&cfg, sizeof cfg).  The function of the same name
 * that the library still exports serves binaries built against ABI <= 3, whose struct had 72 bytes: it fills exactly those. */
void adsb_config_init(adsb_config *cfg, size_t struct_size);
void adsb_config_default(adsb_config *cfg);
#define adsb_config_default(cfg) adsb_config_init((cfg), sizeof(adsb_config))

/* Allocates the stream state that air.c:33-34,49-50 / demod.c:86 / valid.c:30-31
 * keep in statics. NULL on failure (adsb_last_error(NULL) has the reason). */
adsb_decoder *adsb_create(const adsb_config *cfg);
void adsb_destroy(adsb_decoder *d);

/* Restart the stream on the same handle (fresh ring, ts, stats), keeping device buffers
 * and the adsb_profile counters. */
int adsb_reset(adsb_decoder *d);

/* Sample ingress; replaces `decodeiq(const unsigned short *r, const int len)`
 * (air.c:54), called from fileInput (air.c:239) / rx_callback (air.c:175).
 * `samples` is borrowed for the call. Any n is accepted; the stream is the
 * concatenation of all pushes (the reference requires n % 4 == 0).
 * With cfg.push_overlap the call returns once the samples are on the device and the
 * frames follow one call later (link rate at the reference's one-buffer call site). */
int adsb_push(adsb_decoder *d, const uint16_t *samples, size_t n);

/* Overlapped ingress: the same stream semantics as adsb_push, but the call
 * returns as soon as the host-to-device copy and the scan of this chunk are ENQUEUED (copy
 * engine and scan kernel on separate HIP streams); it then collects and resolves the scan
 * of the PREVIOUS chunk, which has been running meanwhile.  Consequences for the caller:
 *   - `samples` stays borrowed until the NEXT adsb_push_async / adsb_push / adsb_finish /
 *     adsb_sync on this handle returns: alternate two buffers, like a double-buffered
 *     read() loop (fileInput's single iqbuff, air.c:230-239, becomes two);
 *   - frames become drainable one call later than with adsb_push (never reordered);
 *   - the buffers should come from adsb_host_alloc() or be adsb_host_register()ed:
 *     pageable memory works but the runtime then copies synchronously (no overlap).
 * adsb_sync() waits for everything in flight: all frames of the samples pushed so far
 * are drainable and every borrowed buffer is free again. */
int adsb_push_async(adsb_decoder *d, const uint16_t *samples, size_t n);
int adsb_sync(adsb_decoder *d);

/* Same, for samples already resident in HBM (device pointer valid on cfg.device).
 * A 16-byte aligned pointer at a stream position that is a multiple of 8 samples
 * is scanned in place; anything else goes through the staging buffer. */
int adsb_push_device(adsb_decoder *d, const void *device_samples, size_t n);

/* adsb_push_device() of the LAST piece of a stream followed by adsb_finish(), in one
 * pass: the in-place scan runs to the exact end of the stream (one launch less, no
 * tail staging).  This is the whole of `adsbdec -f` for a capture that is already
 * resident in HBM. */
int adsb_push_device_final(adsb_decoder *d, const void *device_samples, size_t n);

/* The whole of `adsbdec -f` for ONE capture that is resident in HBM, in one call: adsb_reset + adsb_push_device_final +
 * adsb_take.  Returns the number of frames (*frames as adsb_take), or -1. */
long adsb_decode_device(adsb_decoder *d, const void *device_samples, size_t n, const adsb_frame **frames);

/* Page-locked host buffers for adsb_push(): the counterpart of fileInput's
 * malloc'd iqbuff (air.c:230).  read() straight into one and the push is a single
 * DMA; ordinary malloc'd memory works too, through the driver's bounce buffers. */
void *adsb_host_alloc(size_t bytes);
void adsb_host_free(void *p);
/* Page-lock memory the caller already owns (e.g. buffers a reader thread filled before
 * the GPU runtime was up) so that pushes from it are direct DMA.  0 / -1. */
int adsb_host_register(void *p, size_t bytes);
int adsb_host_unregister(void *p);

/* End of input (fileInput's EOF, air.c:241-244): runs the remaining offsets and
 * applies the reference's end-of-file horizon. */
int adsb_finish(adsb_decoder *d);

/* Frame egress; the records the reference hands to netout() (output.c:159), in
 * the same order. Returns the number copied (<= cap), or -1. */
long adsb_drain(adsb_decoder *d, adsb_frame *out, size_t cap);
/* The same without the copy: every pending frame, in place. *frames points into the
 * handle's own queue and stays valid until the next call that pushes samples into,
 * finishes, resets or destroys this handle; the frames count as drained. Returns their
 * number (0: *frames is NULL). */
long adsb_take(adsb_decoder *d, const adsb_frame **frames);
/* Number of frames currently waiting in the handle. */
size_t adsb_pending(const adsb_decoder *d);

/* print_stats() counters (valid.c:84-100); needs collect_stats=1 for try_, which are counted
 * on the device and fetched by this call (it waits for the count passes still queued). */
int adsb_get_stats(const adsb_decoder *d, adsb_stats *out);
int adsb_get_profile(const adsb_decoder *d, adsb_profile *out);

/* Last error text of a handle, or of the last failed adsb_create() when d==NULL. */
const char *adsb_last_error(const adsb_decoder *d);

/* formatpkt() (output.c:204-262, WITH_AIR). outformat 0 = AVR "*hex;\n",
 * 1 = AVR-MLAT "@ts48hex;\n", 2 = Beast binary. pkt must hold 256 bytes.
 * Returns the packet length. */
int adsb_format_frame(const adsb_frame *f, int outformat, char *pkt);

/* ---- host-side greedy resolver (demod.c:89,99,125-141 + air.c:94-99) --------
 * Exposed so that a host that gathers candidates from several GPUs/ranks can
 * replay the reference's sequential rules once, and so that the
 * host logic is testable without a GPU. */
typedef struct adsb_resolver adsb_resolver;
adsb_resolver *adsb_resolver_create(void);
void adsb_resolver_destroy(adsb_resolver *r);
/* More hands (cfg.host_threads >= 3 in a decoder handle): `helpers` threads (0..15; 0 = none again) write the frames of every
 * adsb_resolver_advance_stream batch that yields at least min_frames frames, the caller only decides them.  Same frames, same
 * order, same counters.  Returns the threads that run, or -1. */
int adsb_resolver_set_threads(adsb_resolver *r, int helpers, size_t min_frames);
/* Candidates (ascending g) and tries (ascending; (g<<2)|code, code 0/1/2 = DF11/17/18),
 * all with g >= the previous g_complete. */
int adsb_resolver_feed(adsb_resolver *r, const adsb_candidate *cands, size_t n_cands,
                       const uint64_t *tries, size_t n_tries);
/* Everything with g < g_complete has been fed; the stream has produced
 * power_samples 10 MS/s samples so far. Appends accepted frames internally. */
int adsb_resolver_advance(adsb_resolver *r, uint64_t power_samples, uint64_t g_complete);
long adsb_resolver_drain(adsb_resolver *r, adsb_frame *out, size_t cap);
int adsb_resolver_stats(const adsb_resolver *r, adsb_stats *out);

/* ---- diagnostics: the device -> host hand-off stream, walked without a device -------------
 * The scan kernel hands its records to the host through ONE stream of 16-byte granules in pinned
 * host memory: per tile a marker {tile, n | flags | lines reserved << 19,
 * check_lo, check_hi} followed by n records of two granules; a tile counts only once its marker's
 * check words agree with the XOR of its record granules, mixed with the launch's `gen`.
 * adsb_handoff_walk applies exactly the rules the streaming collect applies (the same code) to an
 * image of such a stream in ordinary memory: tile_start[t] / tile_count[t] (n_tiles entries each)
 * receive the granule index of tile t's first record and its record count (~0u: not in), the
 * return value is the number of leading tiles that may be handed on (all in, none of them holding),
 * and *status says why the walk ended: 0 every tile is in; 1 a tile holds -- it has records on the
 * loose list (flag 0x10000: the tiles behind it are still read, and wait with it for the launch's end)
 * or its range ran past the array (flag 0x20000: the stream ends there) -- or the stream is full (flag 0x40000, a statistics
 * run's "my tries are on the launch-wide list", holds nothing up);
 * 2 the bytes at the cursor are not (yet) a valid marker of this launch; -1 a tile appears twice.
 * No GPU is needed: this is how the host logic is tested. */
long adsb_handoff_walk(const void *stream, size_t granules, uint32_t n_tiles, uint32_t gen,
                       uint32_t *tile_start, uint32_t *tile_count, int *status);
/* A record of that stream is two granules {g_rel, pw, w0, w1}{w2, w3 | len << 16 | flags << 24, pw', pw''}: the frame's
 * 14 bytes in w0..w3, flags bit 0 = repaired by the 1-bit extension, flags bits 1..2 = the record stands for 1 + that many
 * candidates -- the same bytes decoded at the consecutive offsets g_rel, g_rel + 1 (power pw'), g_rel + 2 (pw''): the
 * half-sample shifted copies of one frame, which stay reachable when frames stand back to back (demod.c:125-141 decides
 * which copy the scan lands on) and would otherwise cost three records per frame on a full channel.
 * adsb_resolver_advance_stream: the streaming collect's hand-over to the resolver over such an image -- every tile must be in
 * (the rules above); the resolver (adsb_resolver_*, in stream or chain mode) then walks the tiles' records where they lie
 * (offsets = g_base + g_rel), like adsb_resolver_feed + adsb_resolver_advance(power_samples, g_complete) would on the
 * expanded candidates; with_head != 0 also copies the head candidates of a chain (adsb_resolver_head).  Returns n_tiles or -1. */
long adsb_resolver_advance_stream(adsb_resolver *r, const void *stream, size_t granules, uint32_t n_tiles, uint32_t gen,
                                  uint64_t g_base, uint64_t power_samples, uint64_t g_complete, int with_head);

/* ---- shard planning -------------------------------------------
 * Splits the offsets [0, power_samples-ADSB_WINDOW] of one stream over n_shards
 * owners. Shard i owns offsets [g_begin[i], g_end[i]) (g_begin multiple of 28) and
 * must be given input samples [first_sample[i], first_sample[i]+n_samples[i]). */
int adsb_plan_shards(uint64_t total_samples, int n_shards, uint64_t *g_begin, uint64_t *g_end,
                     uint64_t *first_sample, uint64_t *n_samples);

/* Scan a stand-alone device buffer that holds stream samples
 * [first_sample, first_sample+n) for the owned offsets [g_begin, g_end) and
 * return its CRC-valid candidates / tries (sorted). No stream state is touched:
 * this is the per-shard call of the multi-GPU path. Returns counts through
 * n_cands/n_tries; -1 on error, -2 if a capacity was too small (counts are set
 * to what is needed). */
int adsb_scan_shard(adsb_decoder *d, const void *device_samples, uint64_t first_sample, size_t n,
                    uint64_t g_begin, uint64_t g_end, adsb_candidate *cands, size_t cand_cap,
                    size_t *n_cands, uint64_t *tries, size_t try_cap, size_t *n_tries);

/* adsb_scan_shard for a (small) window whose samples are in HOST memory: copied to a device buffer of the handle's own,
 * scanned there.  Like adsb_scan_shard it touches neither the handle's stream nor its resolver, so it may be called
 * between adsb_shard_end and the use of the frames that call handed out.  This is how a host-fed shard delivers the two
 * windows of tries adsb_stitch_shards_stats asks for. */
int adsb_scan_shard_host(adsb_decoder *d, const uint16_t *host_samples, uint64_t first_sample, size_t n,
                         uint64_t g_begin, uint64_t g_end, adsb_candidate *cands, size_t cand_cap,
                         size_t *n_cands, uint64_t *tries, size_t try_cap, size_t *n_tries);

/* ---- time-sharded stream, resolved where the records are (BASELINE configs[4]) ---------------
 * adsb_scan_shard + one resolver on one rank funnels every candidate of the stream through a single thread.  The
 * scalable form: every rank resolves its OWN shard while its kernel runs -- the greedy rule of demod.c:89,128,134,141
 * started at the shard's first offset, as if no frame of the previous shard reached into it -- and one rank only
 * repairs the seams, hands out per-shard ts offsets (demod.c:86,99) and applies the end-of-file horizon
 * (air.c:94-99).  No reference counterpart: the reference is one thread on one stream. */
typedef struct adsb_shard_head {
    uint64_t g_begin, g_end; /* the offsets this shard owns                                                      */
    uint64_t n_frames;       /* speculative frames; their ts is LOCAL: g + 1 - (offsets jumped inside the shard) */
    uint64_t n_head;         /* head candidates: EVERY CRC-valid candidate with g < head_end, ascending          */
    uint64_t head_end;
    uint64_t skipped;        /* offsets jumped by the speculative frames: sum of (span - 1)                      */
    uint64_t status;         /* 0 = ok                                                                           */
    uint64_t n_bases;        /* the shard's own walk of the deqframe call chain (adsb_shard_part.bases); 0 = none */
    uint64_t walk_final;     /* 1: that walk ended because the stream does (air.c:94: no further call fires)      */
    uint64_t has_tries;      /* 1: tries[] is filled (the handle was created with collect_stats)                 */
    uint64_t tries[3];       /* valid.c:46,68 for the offsets of this shard as the SPECULATIVE chain visits them: DF-gate
                                passes in [g_begin, g_end) that lie in no speculative frame, per DF (11, 17, 18); counted
                                on the device.  adsb_stitch_shards_stats turns the sum into the stream's Try row. */
    uint64_t ok[3];          /* valid.c:53,75 for the speculative frames, per DF; the stitcher corrects the sum for the   */
    uint64_t fixed;          /* frames a seam repair or the horizon drops and adds (fixed: of them, 1-bit repaired)       */
} adsb_shard_head;

typedef struct adsb_shard_part { /* one shard as the stitcher sees it (plain host pointers, e.g. into shared memory) */
    const adsb_shard_head *head;
    const adsb_frame *frames;
    const adsb_candidate *head_cands;
    const uint64_t *bases;       /* head->n_bases call bases from the guessed entry base g_begin on, or NULL: lets the
                                    stitcher's end-of-file walk jump over the shard once it meets one of them */
    /* Statistics only (adsb_stitch_shards_stats; NULL / 0 otherwise).  The shard's own Try count is right except where
     * the true chain differs from the speculative one: behind a seam, and beyond the end-of-file horizon.  For those
     * two windows the stitcher needs the DF-gate passes themselves, (g << 2) | code ascending, as adsb_scan_shard
     * returns them: */
    const uint64_t *head_tries;  /* EVERY pass with g_begin <= g < head_tries_end (>= min(g_end, head_end + 1200))   */
    uint64_t n_head_tries, head_tries_end;
    const uint64_t *tail_tries;  /* EVERY pass with tail_from <= g < g_end: shards that reach into the stream's last  */
    uint64_t n_tail_tries, tail_from; /* ADSB_TAIL_OFFSETS offsets (tail_from <= the horizon); tail_from = ~0: none    */
} adsb_shard_part;

typedef struct adsb_shard_fix { /* the stitcher's verdict for one shard; its final frames are, in this order,      */
    uint64_t new_first, n_new;  /*   new_frames[new_first .. +n_new): accepted by the seam repair, ts final,       */
    uint64_t drop_front, keep;  /*   frames[drop_front .. +keep):     speculative frames that stand, with          */
    int64_t ts_sub;             /*   ts_final = ts_local - ts_sub (adsb_shard_apply_fix)                           */
} adsb_shard_fix;

/* Scan the owned offsets of a device-resident shard (same buffer rules as adsb_scan_shard) and resolve them on the
 * fly.  frames / head_cands receive at most frame_cap / head_cap entries; -2 if a capacity was too small (head->n_frames
 * / n_head say what is needed).  With collect_stats the shard's own Try count comes back in head->tries.
 * The handle must not hold a stream of its own (fresh or adsb_reset): the call runs the handle's resolver. */
int adsb_scan_shard_resolved(adsb_decoder *d, const void *device_samples, uint64_t first_sample, size_t n,
                             uint64_t g_begin, uint64_t g_end, adsb_shard_head *head, adsb_frame *frames,
                             size_t frame_cap, adsb_candidate *head_cands, size_t head_cap);
/* The same, and the shard's walk of the deqframe call chain (adsb_shard_walk) done on the way, while the kernel runs:
 * total_samples is the whole stream's length; bases / bases_cap as adsb_shard_walk. */
int adsb_scan_shard_resolved_walk(adsb_decoder *d, const void *device_samples, uint64_t first_sample, size_t n,
                                  uint64_t g_begin, uint64_t g_end, uint64_t total_samples, adsb_shard_head *head,
                                  adsb_frame *frames, size_t frame_cap, adsb_candidate *head_cands, size_t head_cap,
                                  uint64_t *bases, size_t bases_cap);
/* adsb_scan_shard_resolved_walk without the copies: the shard's speculative frames and head candidates are handed out IN
 * PLACE (like adsb_shard_end does for a shard fed piecewise) and stay valid until the next call on the handle that scans,
 * pushes or resets.  The call resets the handle first (a stream it held is dropped). */
int adsb_scan_shard_resolved_take(adsb_decoder *d, const void *device_samples, uint64_t first_sample, size_t n,
                                  uint64_t g_begin, uint64_t g_end, uint64_t total_samples, adsb_shard_head *head,
                                  const adsb_frame **frames, const adsb_candidate **head_cands, uint64_t *bases, size_t bases_cap);
/* A shard's own walk of the deqframe call chain over its speculative frames (each rank, in parallel, after its scan):
 * fills bases[0 .. min(cap, n)) and head->n_bases / walk_final; returns n (> cap: too small, n_bases is left 0). */
/* 0 when the caller's adsb_shard_head / adsb_shard_part have the size this library writes and reads (see ADSB_ABI_VERSION 4);
 * -1 otherwise: the caller was built against another layout and must not call the shard API. */
int adsb_shard_layout_check(size_t sizeof_shard_head, size_t sizeof_shard_part);
size_t adsb_shard_walk(adsb_shard_head *head, const adsb_frame *frames, uint64_t total_samples, uint64_t *bases, size_t cap);
/* The serial part, on one rank: parts in shard order.  0; -1 on bad arguments (or inconsistent statistics input); -2 when
 * new_cap is too small (*n_new_total = a lower bound of what is needed: grow new_frames and call again); -3 when a seam cannot
 * be decided from the head candidates (dense overlapping frames through a whole head window): fall back to
 * adsb_scan_shard + adsb_resolver_*. */
int adsb_stitch_shards(const adsb_shard_part *parts, int n_parts, uint64_t total_samples, adsb_shard_fix *fix,
                       adsb_frame *new_frames, size_t new_cap, size_t *n_new_total);
/* The same; walk_stats[0] = calls of the deqframe chain walked here, [1] = calls skipped by jumping onto shards' own walks. */
int adsb_stitch_shards_ex(const adsb_shard_part *parts, int n_parts, uint64_t total_samples, adsb_shard_fix *fix,
                          adsb_frame *new_frames, size_t new_cap, size_t *n_new_total, uint64_t walk_stats[2]);
/* The same, and the stream's statistics (valid.c:84-100) from the shards' own Try counts: `stats->try_` = sum of
 * head->tries, corrected behind every repaired seam and beyond the end-of-file horizon from the parts' head_tries /
 * tail_tries; ok / fixed from the final frames.  -3 also when one of those windows does not cover what the correction
 * needs (the caller falls back to adsb_scan_shard + one resolver, like for an undecidable seam). */
int adsb_stitch_shards_stats(const adsb_shard_part *parts, int n_parts, uint64_t total_samples, adsb_shard_fix *fix,
                             adsb_frame *new_frames, size_t new_cap, size_t *n_new_total, uint64_t walk_stats[2],
                             adsb_stats *stats);
/* Offsets at the end of a stream inside which the end-of-file horizon (air.c:94-99) always lies:
 * every offset below (power samples - ADSB_TAIL_OFFSETS) is visited or jumped whatever the traffic. */
#define ADSB_TAIL_OFFSETS 42181

/* ---- a shard fed PIECEWISE (host-fed multi-GPU path): the handle becomes a stream that starts at sample
 * first_sample and owns the offsets [g_begin, g_end) (adsb_plan_shards).  Between the two calls feed it exactly the
 * plan's samples with adsb_push / adsb_push_async / adsb_push_device -- copy and scan of successive pieces overlap as
 * for any stream -- and it is resolved in chain mode on the fly, like adsb_scan_shard_resolved_walk does for a buffer
 * that is resident in HBM.  adsb_shard_end hands the shard's speculative frames and head candidates out IN PLACE: the
 * pointers stay valid until the next adsb_reset / adsb_shard_begin / adsb_destroy of the handle.  With collect_stats
 * the shard's own Try count is in head->tries.  bases / bases_cap as adsb_scan_shard_resolved_walk (may be NULL / 0). */
int adsb_shard_begin(adsb_decoder *d, uint64_t first_sample, uint64_t g_begin, uint64_t g_end, uint64_t total_samples,
                     uint64_t *bases, size_t bases_cap);
int adsb_shard_end(adsb_decoder *d, adsb_shard_head *head, const adsb_frame **frames, const adsb_candidate **head_cands);

/* ts_final = ts_local - ts_sub, in place, for frames[0 .. n). */
void adsb_shard_apply_fix(adsb_frame *frames, size_t n, int64_t ts_sub);
/* The host-side resolver in the same chain mode (tests; hosts that hold candidates themselves): call before the first
 * feed.  adsb_resolver_head copies the head candidates out; adsb_resolver_skipped is adsb_shard_head.skipped. */
int adsb_resolver_start_chain(adsb_resolver *r, uint64_t g_begin, uint64_t head_end);
/* ... with the shard's walk of the deqframe calls advanced beside it (what adsb_scan_shard_resolved_walk does): call right
 * after adsb_resolver_start_chain; bases must stay valid until the last adsb_resolver_advance.  adsb_resolver_walk_result:
 * the number of bases (as adsb_shard_walk returns it) and *final. */
int adsb_resolver_start_walk(adsb_resolver *r, uint64_t g_begin, uint64_t g_end, uint64_t total_samples, uint64_t *bases, size_t cap);
size_t adsb_resolver_walk_result(const adsb_resolver *r, int *final);
long adsb_resolver_head(adsb_resolver *r, adsb_candidate *out, size_t cap);
uint64_t adsb_resolver_skipped(const adsb_resolver *r);

/* The CPUs that are local to HIP device `device` (the GPU's NUMA node), as the kernel prints them
 * (/sys/bus/pci/devices/<bdf>/local_cpulist, e.g. "0-63,128-191"), into out.  Why a host wants it: the thread that feeds a
 * handle polls memory the device writes (the hand-off stream); on the far socket of a two-socket
 * host that thread was measured 2.5-3 x slower.  adsb_multi_create binds its workers with it; a host with threads of its
 * own does the same for the thread that calls adsb_push*.  Returns the string's length, 0 when the platform does not say
 * (numa_node = -1), -1 on error. */
int adsb_device_cpulist(int device, char *out, size_t cap);
/* The NUMA node HIP device `device` hangs off (/sys/bus/pci/devices/<bdf>/numa_node); -1 when the platform does not say. */
int adsb_device_numa_node(int device);

/* ---- where a host-resident capture lives (csrc/numa.cpp) ---------------------------------------------------------------
 * Stands where the reference has `iqbuff = malloc(...)` (air.c:230).  A capture in host memory that eight devices pull at
 * once, each over its own link, should have every slice on the socket its device hangs off: a slice on the other socket
 * crosses the socket fabric, which four of the eight links then share.
 * adsb_host_alloc_on: page-locked memory (2 MiB-aligned, huge pages advised) bound to the node of `device` with mbind(),
 *   first-touched there, then registered with the runtime.  Best effort: where there is one node, or the policy call is
 *   refused, the memory is page-locked where the kernel put it.  NULL when it cannot be mapped or page-locked.
 * adsb_host_alloc_sharded: ONE array of total_samples samples for a capture that will be decoded in shards (first_sample /
 *   n_samples as adsb_plan_shards / adsb_multi_plan give them), shard i's part on the node of devices[i]; the boundary
 *   between two nodes lies where the next shard starts, rounded to a huge page.
 * Both are freed with adsb_host_free.
 * adsb_host_placement: on which node do the pages of [p, p + bytes) live?  Samples up to 256 pages with move_pages():
 *   *major_node = the node most of them are on, *fraction_on_want = the share on want_node.  -1 when the kernel does not say. */
void *adsb_host_alloc_on(size_t bytes, int device);
uint16_t *adsb_host_alloc_sharded(uint64_t total_samples, int n_shards, const uint64_t *first_sample, const uint64_t *n_samples,
                                  const int *devices);
int adsb_host_placement(const void *p, size_t bytes, int want_node, int *major_node, double *fraction_on_want);
int adsb_host_release_mapped(void *p); /* adsb_host_free's first look (1: p was a mapping of the two calls above and is gone) */

/* ---- ONE process, several GPUs (csrc/multi.cpp) ---------------------------------------------------------------
 * The host of BASELINE configs[3] / configs[4]: a worker thread and a decoder handle per device; no collective on the
 * data path.  Stands where fileInput's loop (air.c:217-246) hands its buffers to decodeiq and the frames
 * come back in ascending order for netout (output.c:159-182). */
typedef struct adsb_multi adsb_multi;

typedef struct adsb_multi_info { /* of the last adsb_multi_decode_* call */
    int32_t shards;        /* shards the capture was cut into (streams decoded side by side, for the stream calls)     */
    int32_t fallback;      /* 1: a seam could not be decided from the head candidates; the capture then went through
                              ONE device as an ordinary stream (same frames)                                         */
    uint64_t calls_walked; /* deqframe calls the end-of-file walk replayed on the calling thread ...                   */
    uint64_t calls_jumped; /* ... and skipped by jumping onto the shards' own walks                                    */
    double create_ms;      /* adsb_multi_create: the slowest worker's adsb_create (the devices start side by side)     */
    double workers_ms;     /* the slowest worker's share of the call                                                  */
    double stitch_us;      /* adsb_stitch_shards[_stats] on the calling thread                                        */
    double serial_us;      /* everything behind the last worker: stitch + gather into one array                       */
    double total_ms;
    int32_t workers_bound; /* workers whose thread runs on the CPUs of its device's NUMA node (adsb_device_cpulist)       */
    int32_t reserved;
} adsb_multi_info;

/* n_devices workers; devices[i] = HIP ordinal of worker i (NULL: 0 .. n_devices-1).  An ordinal may repeat: several
 * handles on one device (plumbing tests on a one-GPU box).  cfg as for adsb_create (device and stream are ignored);
 * each worker creates its own handle, so the devices' runtimes come up in parallel.  NULL on failure
 * (adsb_multi_last_error(NULL)). */
adsb_multi *adsb_multi_create(const adsb_config *cfg, int n_devices, const int *devices);
void adsb_multi_destroy(adsb_multi *m);
int adsb_multi_devices(const adsb_multi *m);

/* configs[4]: ONE capture, time-sharded: adsb_plan_shards over as many devices as the capture is worth (at least 128 Ki
 * offsets per shard), each worker feeds its halo'd slice to its device in 32 MiB pieces -- the copy of a piece overlaps
 * the scan of the one before, the shard is resolved while its kernels run -- and the calling thread stitches
 * (adsb_stitch_shards, adsb_shard_apply_fix) and gathers.  Returns the number of frames, in the reference's order, *frames
 * valid until the next call on m; -1 on failure.  With cfg.collect_stats the stream's Try/Ok table is available from
 * adsb_multi_get_stats afterwards.
 *   _host:   the capture lies in host memory (page-lock it -- adsb_host_register / adsb_host_alloc -- or every piece goes
 *            through the runtime's bounce buffers);
 *   _file:   every worker reads its own slice of a regular file (pread) into page-locked buffers of its own;
 *   _device: slice i is resident in the HBM of worker i's device and holds the samples adsb_multi_plan says (16-byte
 *            aligned); n_slices must be the plan's shard count. */
long adsb_multi_decode_host(adsb_multi *m, const uint16_t *samples, size_t n, const adsb_frame **frames);
long adsb_multi_decode_file(adsb_multi *m, const char *path, const adsb_frame **frames);
long adsb_multi_decode_device(adsb_multi *m, uint64_t total_samples, const void *const *slices, int n_slices,
                              const adsb_frame **frames);
/* The plan those calls use for a stream of total_samples: fills the first <return value> entries (arrays of
 * adsb_multi_devices(m) entries), as adsb_plan_shards does. */
int adsb_multi_plan(const adsb_multi *m, uint64_t total_samples, uint64_t *g_begin, uint64_t *g_end,
                    uint64_t *first_sample, uint64_t *n_samples);
int adsb_multi_get_stats(const adsb_multi *m, adsb_stats *out);

/* configs[3]: n_streams INDEPENDENT captures, stream s on worker s mod adsb_multi_devices(m), each with its own ts and
 * statistics -- N times what `adsbdec -f` does, side by side.  0 / -1; results per stream afterwards. */
int adsb_multi_decode_streams_host(adsb_multi *m, int n_streams, const uint16_t *const *samples, const size_t *n);
int adsb_multi_decode_streams_file(adsb_multi *m, int n_streams, const char *const *paths);
long adsb_multi_stream_frames(const adsb_multi *m, int stream, const adsb_frame **frames);
int adsb_multi_stream_stats(const adsb_multi *m, int stream, adsb_stats *out);

int adsb_multi_get_info(const adsb_multi *m, adsb_multi_info *out);
/* A page-locked array for ONE capture of total_samples samples that adsb_multi_decode_host will decode: laid out with
 * adsb_host_alloc_sharded over m's own plan and devices, so that every worker pulls its slice from its own socket.  Free
 * with adsb_host_free.  adsb_multi_worker_placement says, per worker, whether that held for the last host-fed decode. */
uint16_t *adsb_multi_host_alloc(adsb_multi *m, uint64_t total_samples);
typedef struct adsb_worker_placement {
    int32_t device, device_node; /* the worker's HIP device and its NUMA node (-1: the platform does not say)            */
    int32_t thread_bound;        /* the worker's thread runs on the CPUs of that node                                    */
    int32_t slice_node;          /* the node most pages of the worker's slice of the last adsb_multi_decode_host capture
                                    live on; -1: not a host source, or the kernel does not say                           */
    double local_fraction;       /* share of the sampled pages of that slice on device_node (1.0 = the link is fed from
                                    its own socket)                                                                      */
} adsb_worker_placement;
int adsb_multi_worker_placement(const adsb_multi *m, int worker, adsb_worker_placement *out);
/* adsb_get_profile of worker `worker`'s handle (launches, kernel time on the device clock with cfg.profile ...). */
int adsb_multi_worker_profile(const adsb_multi *m, int worker, adsb_profile *out);
/* Last error text of m, or of the last failed adsb_multi_create() when m == NULL; names the device and worker. */
const char *adsb_multi_last_error(const adsb_multi *m);

int adsb_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
