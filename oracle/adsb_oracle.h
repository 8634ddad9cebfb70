/*
 * adsb_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, strict IEEE-754 binary32, no FMA contraction) of the
 * TLeconte/adsbdec offline "-f" demodulation path.  It exists so that tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg can check / time the
 * HIP path against the reference's algorithm.  Nothing under adsbdec_amd/ (the
 * product) may include, link, import or execute anything in oracle/.
 *
 * Parity pinning status (see DESIGN.md "Oracle"): PINNED, every stage, against the
 * reference's own code EXECUTED here.  oracle/Makefile compiles air.c:29-101
 * (fbuff/fidx, dsfilter, ampbuff/aidx, decodeiq -- the part of air.c that uses no
 * libairspy symbol), demod.c, valid.c (+ crc.h) and output.c:formatpkt from
 * /root/reference into oracle/_ref/; tests/test_oracle_vs_ref.py requires this
 * restatement to agree with that chain bit for bit (power samples, frames, ts, pw,
 * Try/Ok, AVR/MLAT/Beast bytes) over > 2000 seeded uint16 captures, and the
 * fixtures in tests/golden/ are minted through it (oracle/make_golden.py).
 * The 1-bit repair EXTENSION (orc_decode_fix1) has no reference counterpart and
 * stays "parity unpinned" by nature.
 */
#ifndef ADSB_ORACLE_H
#define ADSB_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_PULSEW 5                        /* adsbdec.h:1 */
#define ORC_DECOFFSET (240 * ORC_PULSEW)    /* adsbdec.h:3 */
#define ORC_FLTLEN 14                       /* air.c:32   */
#define ORC_APBUFFSZ (8196 * ORC_PULSEW)    /* air.c:47   */

/* Record leaving the path: mirrors blk_t / netout() arguments (output.c:45-52,159). */
typedef struct {
    uint64_t g;        /* global power-sample index of the preamble (not in the reference; bookkeeping) */
    uint64_t ts;       /* demod.c:86,99 loop-iteration counter at acceptance */
    uint32_t pw;       /* (p1+p2)/4, demod.c:127,133 */
    uint8_t len;       /* 7 or 14 */
    uint8_t frame[14];
} orc_frame_t;

typedef void (*orc_sink_fn)(void *user, const orc_frame_t *f);

typedef struct {
    /* air.c:33-34 */
    float ring[ORC_FLTLEN];
    uint32_t fidx;
    /* air.c:49-50 */
    float ampbuff[ORC_APBUFFSZ + 4];
    uint32_t aidx;
    /* demod.c:26,86 */
    int df;
    /* EXTENSION, not in the reference: accept DF17/18 frames whose CRC
     * residual is the syndrome of exactly one bit k in [5,112) after flipping it.
     * No reference parity exists for this mode. */
    int fix1;
    uint32_t stat_fixed;
    uint64_t ts;
    /* valid.c:30-31 */
    uint32_t stat_try[32];
    uint32_t stat_ok[32];
    /* bookkeeping only */
    uint64_t gbase;      /* global power index of ampbuff[0] */
    uint64_t n_deq_calls;
    orc_sink_fn sink;
    void *sink_user;
} orc_state_t;

void orc_init(orc_state_t *o, int df18, orc_sink_fn sink, void *user);

/* air.c:54-101.  len is rounded UP to a multiple of 4; the pad samples are 2048
 * (the reference over-reads stale buffer bytes there; those power samples can
 * never be visited by deqframe -- DESIGN.md "partial quad"). */
void orc_decodeiq(orc_state_t *o, const uint16_t *r, size_t len);

/* demod.c:84-144 */
int orc_deqframe(orc_state_t *o, const float *amp, int len);

/* air.c:217-246: feeds orc_decodeiq in IQBUFFSZ (1 Mi sample) chunks. */
void orc_decode_buffer(orc_state_t *o, const uint16_t *x, size_t n);

/* Convenience: whole-buffer decode into a caller array. Returns frame count
 * (may exceed cap; only cap are stored). stats may be NULL: else try[3],ok[3] for DF11/17/18. */
size_t orc_decode(const uint16_t *x, size_t n, int df18, orc_frame_t *out, size_t cap,
                  uint32_t *stats6);
/* Same with the 1-bit correction EXTENSION enabled (parity unpinned vs the reference). */
size_t orc_decode_fix1(const uint16_t *x, size_t n, int df18, orc_frame_t *out, size_t cap,
                       uint32_t *stats6, uint32_t *n_fixed);

/* demod.c/valid.c stages only, driven on power samples (see ref_harness.c). */
size_t orc_demod_power(const float *a, size_t m, int df18, orc_frame_t *out, size_t cap,
                       uint32_t *stats6);

/* Front end only: power samples a[m], m < 2*ceil(n/4), ring zero-initialised. */
size_t orc_power(const uint16_t *x, size_t n, float *a);

/* crc.h:36-42 on a 7/14 byte frame; returns the 24-bit residual. */
uint32_t orc_crc_residual(const uint8_t *frame, int n);
/* crc.h:1-34 table entry (generated from 0xFFF409, MSB first). */
uint32_t orc_crc_table(int i);

/* Stateless per-offset evaluation used to test candidate-based designs:
 * returns 0 if the preamble test fails, 1 if it passes but the DF gate rejects,
 * 2 if DF gate passes but CRC fails, 3 if CRC-valid. frame/len/pw filled for >=2. */
int orc_eval_offset(const float *a, int df18, uint8_t frame[14], int *len, uint32_t *pw);

/* Every offset in [g0,g1) evaluated independently: CRC-valid ones go to cands
 * (ts unused), DF-gate passes to tries as (g<<2)|code. Returns the try count. */
size_t orc_scan_all(const float *a, uint64_t g0, uint64_t g1, int df18, orc_frame_t *cands,
                    size_t cand_cap, size_t *n_cands, uint64_t *tries, size_t try_cap);

/* output.c:204-262 with WITH_AIR. outformat 0 AVR, 1 AVR-MLAT, 2 Beast. */
int orc_formatpkt(const uint8_t *frame, int len, uint64_t ts, uint32_t pw, int outformat,
                  char *pkt);

#ifdef __cplusplus
}
#endif
#endif
