/*
 * adsbdec_amd_diag.h -- what lies UNDER the calls of adsbdec_amd.h: the primitives the multi-GPU driver is built from, the
 * host-side resolver and the hand-off walker as handles of their own (how the host logic is tested without a GPU), the
 * NUMA placement queries, and the test knobs (adsb_debug_config).  A drop-in host needs none of it: the patch of
 * INTEGRATION.md and the C host program build against adsbdec_amd.h alone (tests/test_gpu_dropin.py checks that).
 * Same library, same ABI version; every entry point names the reference rule it replays (file:line under /root/reference).
 */
#ifndef ADSBDEC_AMD_DIAG_H
#define ADSBDEC_AMD_DIAG_H

#include "adsbdec_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Test knobs (0 = default): adsb_config.debug points at one of these; adsb_create copies it.  The library reads no
 * environment variable: whatever a test has to force is here.  Grows at its end (struct_size, like adsb_config). */
typedef struct adsb_debug_config {
    uint32_t struct_size;     /* sizeof(adsb_debug_config) as the caller knows it                                      */
    int32_t queue_cap;        /* survivor-queue entries per workgroup round (256..1024)                               */
    int32_t cand_cap;         /* loose-list records per launch slot: forces the relaunch-with-regrown-buffers path    */
    int32_t try_cap;          /* try words per launch slot (collect_stats); every try then goes to the launch-wide list */
    int32_t clist_cap;        /* CRC-valid candidates staged per tile (1..256)                                        */
    int32_t no_streaming;     /* 1: every launch is collected after completion (no hand-off stream)                   */
    int32_t frames_cap;       /* collect_stats: accepted frames the upload buffers start with (regrow path)           */
    int32_t reader_min_tiles; /* host_threads >= 2: launches of at least this many tiles go through the reader thread */
    int32_t shard_head;       /* resolved shards: offsets whose candidates are all kept for the stitcher (ADSB_SHARD_HEAD) */
    int32_t passes;           /* passes per tile of every launch (2..32) instead of the cost model's choice           */
    int32_t big_tiles;        /* tiles of a launch that take all its passes; the rest four (scan_kernel.h tile_passes) */
    int32_t gang_min;         /* host_threads >= 3: batches of at least this many records go through the gang
                                 (default: 2048 records decided ahead, 1024 frames written)                           */
} adsb_debug_config;

/* A CRC-valid candidate before greedy resolution (what one GPU emits for the
 * offsets it owns; the unit the host gathers across shards). */
typedef struct adsb_candidate {
    uint64_t g;
    uint32_t pw;
    uint8_t len;
    uint8_t frame[14];
    uint8_t reserved;
} adsb_candidate;

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
/* The head window of a shard: every CRC-valid candidate of its first ADSB_SHARD_HEAD offsets is kept for the stitcher, which
 * re-runs the greedy chain from the true entry position over them until it accepts a candidate the shard's speculative chain
 * accepted too.  16 384 until round 6; on a channel whose frames stand back to back (BASELINE configs[2]) two chains that
 * entered a run of frames at different half-sample copies stay apart until a copy is missing or the run ends -- hundreds of
 * frames -- and every seam of such a capture was undecidable (-3: the driver then decodes on ONE device).  262 144 offsets are
 * 218 frame lengths; the cost is a few hundred candidates and one small stateless scan per shard. */
#define ADSB_SHARD_HEAD 262144
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
/* NULL when this host can run the library (its host side is built for x86-64 with AVX2), else the message adsb_create fails
 * with (adsb_last_error(NULL)).  The function itself is built for the base instruction set. */
const char *adsb_host_cpu_refusal(void);
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
uint16_t *adsb_host_alloc_sharded(uint64_t total_samples, int n_shards, const uint64_t *first_sample, const uint64_t *n_samples,
                                  const int *devices);
int adsb_host_placement(const void *p, size_t bytes, int want_node, int *major_node, double *fraction_on_want);
int adsb_host_release_mapped(void *p); /* adsb_host_free's first look (1: p was a mapping of the two calls above and is gone) */

#ifdef __cplusplus
}
#endif
#endif
