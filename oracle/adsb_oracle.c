/*
 * adsb_oracle.c -- TEST INFRASTRUCTURE ONLY (see adsb_oracle.h for the rules and
 * the parity-pinning status).
 *
 * A sequential, single-threaded restatement of the reference "-f" path:
 *   air.c:54-101   sample conditioning, fs/4 mix, 14-tap FIR, power, carry
 *   demod.c:31-144 preamble test, DF gate, PPM slicer, greedy advance, ts
 *   crc.h:36-42    CRC-24 (generator 0xFFF409)
 *   valid.c:39-82  residual==0 gate + Try/Ok counters
 *   output.c:204-262 AVR / AVR-MLAT / Beast formatting
 *
 * Build with: gcc -O2 -ffp-contract=off  (binary32 mul then add, never fused).
 */
#include "adsb_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* air.c:36-45.  The reference initialises a float array from double literals, so
 * each tap is (float)<double literal>; written the same way here.  The table is
 * doubled so that index k+o (o in 2..14) needs no modulo. */
static const double k_tap_literals[ORC_FLTLEN] = {
    0.012627, 0.025254, 0.037881, 0.050508, 0.063135, 0.075761, 0.088388,
    0.088388, 0.075761, 0.063135, 0.050508, 0.037881, 0.025254, 0.012627,
};
static float g_taps2[2 * ORC_FLTLEN];
static uint32_t g_crc[256];
static int g_tables_ready;

static void build_tables(void)
{
    if (g_tables_ready)
        return;
    for (int j = 0; j < 2 * ORC_FLTLEN; j++)
        g_taps2[j] = (float)k_tap_literals[j % ORC_FLTLEN];
    /* crc.h:1-34 is the MSB-first byte table of the Mode-S generator 0xFFF409. */
    for (int b = 0; b < 256; b++) {
        uint32_t c = (uint32_t)b << 16;
        for (int k = 0; k < 8; k++)
            c = (c & 0x800000u) ? ((c << 1) ^ 0xFFF409u) : (c << 1);
        g_crc[b] = c & 0xFFFFFFu;
    }
    g_tables_ready = 1;
}

uint32_t orc_crc_table(int i)
{
    build_tables();
    return g_crc[i & 255];
}

void orc_init(orc_state_t *o, int df18, orc_sink_fn sink, void *user)
{
    build_tables();
    memset(o, 0, sizeof *o);
    o->df = df18 ? 1 : 0;
    o->sink = sink;
    o->sink_user = user;
}

/* ---- crc.h:36-42 -------------------------------------------------------- */
static uint32_t crc_step(uint8_t in, uint32_t crc)
{
    return (crc << 8) ^ g_crc[in ^ (uint8_t)(crc >> 16)];
}

uint32_t orc_crc_residual(const uint8_t *frame, int n)
{
    uint32_t crc = 0;
    build_tables();
    for (int k = 0; k < n - 3; k++) /* valid.c:49-50 / 71-72 */
        crc = crc_step(frame[k], crc);
    uint32_t tail = ((uint32_t)frame[n - 3] << 16) | ((uint32_t)frame[n - 2] << 8) | frame[n - 1];
    return (crc & 0xFFFFFFu) ^ tail; /* crc.h:40-42 */
}

/* ---- demod.c:31-81 ------------------------------------------------------ */
static uint8_t slice_byte(const float *a, int idx) /* getabyte, demod.c:31-44 */
{
    uint8_t b = 0;
    for (int i = 0; i < 8; i++)
        if (a[idx + 2 * i * ORC_PULSEW] > a[idx + (2 * i + 1) * ORC_PULSEW])
            b |= (uint8_t)(0x80 >> i);
    return b;
}

/* getdf, demod.c:46-81: a decision tree over the top five bits that accepts
 * exactly 01011 (DF11, 7 bytes), 10001 (DF17, 14) and, when df!=0, 10010 (DF18,
 * 14); the low three bits are sliced unconditionally.  A rejected byte is
 * reported as 0 (demod.c:113-116).  Checked against the real getdf through
 * oracle/_ref (tests/test_oracle_vs_ref.py). */
static uint8_t slice_df(const float *a, int idx, int df, int *len)
{
    uint8_t b = slice_byte(a, idx);
    switch (b >> 3) {
    case 11:
        *len = 7;
        return b;
    case 17:
        *len = 14;
        return b;
    case 18:
        if (!df)
            return 0;
        *len = 14;
        return b;
    default:
        return 0;
    }
}

/* EXTENSION (not in the reference): single-bit repair of long frames.  The residual
 * of a frame with one flipped bit k is x^(111-k) mod G; bits 0..4 (the DF field
 * that selected the frame) are never touched. */
static int try_fix1(uint8_t *frame, uint32_t residual)
{
    uint32_t r = 1;
    for (int k = 111; k >= 5; k--) {
        if (r == residual) {
            frame[k >> 3] ^= (uint8_t)(0x80 >> (k & 7));
            return 1;
        }
        r <<= 1;
        if (r & 0x1000000u)
            r ^= 0x1FFF409u;
    }
    return 0;
}

static int valid_frame(orc_state_t *o, uint8_t *frame, int n, uint64_t ts, uint32_t pw,
                       uint64_t g)
{
    int type = frame[0] >> 3;
    o->stat_try[type]++; /* valid.c:46,68 */
    uint32_t residual = orc_crc_residual(frame, n);
    if (residual != 0) {
        if (!(o->fix1 && n == 14 && try_fix1(frame, residual)))
            return 0;
        o->stat_fixed++;
    }
    o->stat_ok[type]++; /* valid.c:53,75 */
    if (o->sink) {
        orc_frame_t f;
        memset(&f, 0, sizeof f);
        f.g = g;
        f.ts = ts;
        f.pw = pw;
        f.len = (uint8_t)n;
        memcpy(f.frame, frame, (size_t)n);
        o->sink(o->sink_user, &f); /* netout, valid.c:54,76 */
    }
    return 1;
}

/* ---- demod.c:84-144 ----------------------------------------------------- */
int orc_deqframe(orc_state_t *o, const float *amp, int len)
{
    int idx = 0;
    o->n_deq_calls++;
    while (idx < len - ORC_DECOFFSET) {
        o->ts++; /* demod.c:99 */
        /* demod.c:102-105: float add, then C conversion to int (truncation) */
        int p1 = amp[idx] + amp[idx + 2 * ORC_PULSEW];
        int s1 = amp[idx + ORC_PULSEW] + amp[idx + 3 * ORC_PULSEW];
        int p2 = amp[idx + 7 * ORC_PULSEW] + amp[idx + 9 * ORC_PULSEW];
        int s2 = amp[idx + 6 * ORC_PULSEW] + amp[idx + 8 * ORC_PULSEW];
        if (p1 > 2 * s1 && p2 > 2 * s2) { /* SN 2, demod.c:83,107 */
            uint8_t frame[14];
            int nbytes = 0;
            int l = idx + 16 * ORC_PULSEW;
            frame[0] = slice_df(amp, l, o->df, &nbytes);
            if (frame[0] != 0) {
                for (int k = 1; k < nbytes; k++) {
                    l += 16 * ORC_PULSEW;
                    frame[k] = slice_byte(amp, l);
                }
                l += 16 * ORC_PULSEW;
                if (valid_frame(o, frame, nbytes, o->ts, (uint32_t)((p1 + p2) / 4),
                                o->gbase + (uint64_t)idx)) {
                    idx = l; /* demod.c:128,134: jump past the frame */
                    continue;
                }
            }
        }
        idx++;
    }
    return idx;
}

/* ---- air.c:54-101 ------------------------------------------------------- */
static void push_pair_and_filter(orc_state_t *o, float v0, float v1)
{
    o->ring[o->fidx % ORC_FLTLEN] = v0;
    o->fidx++;
    o->ring[o->fidx % ORC_FLTLEN] = v1;
    o->fidx++;

    float si = 0, sq = 0;
    int off = ORC_FLTLEN - (int)(o->fidx % ORC_FLTLEN);
    for (int k = 0; k < ORC_FLTLEN; k += 2) { /* physical ring order, air.c:71-74 */
        si += g_taps2[k + off] * o->ring[k];
        sq += g_taps2[k + 1 + off] * o->ring[k + 1];
    }
    o->ampbuff[o->aidx++] = si * si + sq * sq;
}

static void carry_if_full(orc_state_t *o)
{
    if (o->aidx >= ORC_APBUFFSZ) { /* air.c:94-99 */
        int used = orc_deqframe(o, o->ampbuff, (int)o->aidx);
        if ((uint32_t)used < o->aidx)
            memmove(o->ampbuff, o->ampbuff + used, (o->aidx - (uint32_t)used) * sizeof(float));
        o->aidx -= (uint32_t)used;
        o->gbase += (uint64_t)used;
    }
}

void orc_decodeiq(orc_state_t *o, const uint16_t *r, size_t len)
{
    for (size_t i = 0; i < len; i += 4) {
        float v[4];
        for (int k = 0; k < 4; k++) {
            uint16_t s = (i + (size_t)k < len) ? r[i + (size_t)k] : 0x800;
            v[k] = (float)s - 0x800; /* air.c:64,66,79,81 */
        }
        push_pair_and_filter(o, v[0], v[1]);   /* air.c:64-77 */
        push_pair_and_filter(o, -v[2], -v[3]); /* air.c:79-92: fs/4 sign flip */
        carry_if_full(o);
    }
}

void orc_decode_buffer(orc_state_t *o, const uint16_t *x, size_t n)
{
    const size_t chunk = 1024 * 1024; /* IQBUFFSZ, air.c:218 */
    for (size_t pos = 0; pos < n; pos += chunk) {
        size_t m = n - pos < chunk ? n - pos : chunk;
        orc_decodeiq(o, x + pos, m);
    }
    /* EOF: the partially filled ampbuff is discarded (air.c:241-244). */
}

typedef struct {
    orc_frame_t *out;
    size_t cap, n;
} collect_t;

static void collect_sink(void *user, const orc_frame_t *f)
{
    collect_t *c = (collect_t *)user;
    if (c->n < c->cap)
        c->out[c->n] = *f;
    c->n++;
}

size_t orc_decode(const uint16_t *x, size_t n, int df18, orc_frame_t *out, size_t cap,
                  uint32_t *stats6)
{
    return orc_decode_fix1(x, n, df18 | 0x100, out, cap, stats6, NULL);
}

size_t orc_decode_fix1(const uint16_t *x, size_t n, int df18, orc_frame_t *out, size_t cap,
                       uint32_t *stats6, uint32_t *n_fixed)
{
    orc_state_t *o = (orc_state_t *)malloc(sizeof *o);
    collect_t c = {out, cap, 0};
    orc_init(o, df18 & 0xFF, collect_sink, &c);
    o->fix1 = (df18 & 0x100) ? 0 : 1; /* 0x100 = plain reference behaviour (called from orc_decode) */
    orc_decode_buffer(o, x, n);
    if (n_fixed)
        *n_fixed = o->stat_fixed;
    if (stats6) {
        stats6[0] = o->stat_try[11];
        stats6[1] = o->stat_try[17];
        stats6[2] = o->stat_try[18];
        stats6[3] = o->stat_ok[11];
        stats6[4] = o->stat_ok[17];
        stats6[5] = o->stat_ok[18];
    }
    free(o);
    return c.n;
}

size_t orc_power(const uint16_t *x, size_t n, float *a)
{
    /* Same arithmetic as orc_decodeiq, without the 40980-sample carry. */
    orc_state_t *o = (orc_state_t *)malloc(sizeof *o);
    size_t m = 0;
    orc_init(o, 0, NULL, NULL);
    for (size_t i = 0; i < n; i += 4) {
        float v[4];
        for (int k = 0; k < 4; k++) {
            uint16_t s = (i + (size_t)k < n) ? x[i + (size_t)k] : 0x800;
            v[k] = (float)s - 0x800;
        }
        o->aidx = 0;
        push_pair_and_filter(o, v[0], v[1]);
        push_pair_and_filter(o, -v[2], -v[3]);
        a[m++] = o->ampbuff[0];
        a[m++] = o->ampbuff[1];
    }
    free(o);
    return m;
}

int orc_eval_offset(const float *a, int df18, uint8_t frame[14], int *len, uint32_t *pw)
{
    build_tables();
    int p1 = a[0] + a[2 * ORC_PULSEW];
    int s1 = a[ORC_PULSEW] + a[3 * ORC_PULSEW];
    int p2 = a[7 * ORC_PULSEW] + a[9 * ORC_PULSEW];
    int s2 = a[6 * ORC_PULSEW] + a[8 * ORC_PULSEW];
    if (!(p1 > 2 * s1 && p2 > 2 * s2))
        return 0;
    int nbytes = 0;
    int l = 16 * ORC_PULSEW;
    frame[0] = slice_df(a, l, df18 ? 1 : 0, &nbytes);
    if (frame[0] == 0)
        return 1;
    for (int k = 1; k < nbytes; k++) {
        l += 16 * ORC_PULSEW;
        frame[k] = slice_byte(a, l);
    }
    *len = nbytes;
    *pw = (uint32_t)((p1 + p2) / 4);
    return orc_crc_residual(frame, nbytes) == 0 ? 3 : 2;
}

/* demod/valid stages only, on a caller-supplied power array: replays air.c:94-99's
 * accumulate/carry around orc_deqframe exactly as oracle/ref_harness.c does
 * around the real deqframe. */
size_t orc_demod_power(const float *a, size_t m, int df18, orc_frame_t *out, size_t cap,
                       uint32_t *stats6)
{
    orc_state_t *o = (orc_state_t *)malloc(sizeof *o);
    collect_t c = {out, cap, 0};
    orc_init(o, df18, collect_sink, &c);
    for (size_t i = 0; i + 1 < m; i += 2) {
        o->ampbuff[o->aidx++] = a[i];
        o->ampbuff[o->aidx++] = a[i + 1];
        carry_if_full(o);
    }
    if (stats6) {
        stats6[0] = o->stat_try[11];
        stats6[1] = o->stat_try[17];
        stats6[2] = o->stat_try[18];
        stats6[3] = o->stat_ok[11];
        stats6[4] = o->stat_ok[17];
        stats6[5] = o->stat_ok[18];
    }
    free(o);
    return c.n;
}

/* Exhaustive, order-free evaluation of every offset g in [g0, g1) of a power
 * array (what a data-parallel device computes); used to test the candidate
 * resolver and the shard planner without a GPU. */
size_t orc_scan_all(const float *a, uint64_t g0, uint64_t g1, int df18, orc_frame_t *cands,
                    size_t cand_cap, size_t *n_cands, uint64_t *tries, size_t try_cap)
{
    size_t nc = 0, nt = 0;
    for (uint64_t g = g0; g < g1; g++) {
        uint8_t fr[14];
        int len = 0;
        uint32_t pw = 0;
        int k = orc_eval_offset(a + g, df18, fr, &len, &pw);
        if (k < 2)
            continue;
        if (nt < try_cap)
            tries[nt] = (g << 2) | (uint64_t)((fr[0] >> 3) == 11 ? 0 : (fr[0] >> 3) == 17 ? 1 : 2);
        nt++;
        if (k == 3) {
            if (nc < cand_cap) {
                memset(&cands[nc], 0, sizeof cands[nc]);
                cands[nc].g = g;
                cands[nc].pw = pw;
                cands[nc].len = (uint8_t)len;
                memcpy(cands[nc].frame, fr, (size_t)len);
            }
            nc++;
        }
    }
    *n_cands = nc;
    return nt;
}

/* ---- output.c:204-262 (WITH_AIR) ---------------------------------------- */
int orc_formatpkt(const uint8_t *frame, int len, uint64_t ts, uint32_t pw, int outformat,
                  char *pkt)
{
    static const char hex[] = "0123456789ABCDEF";
    uint64_t ts12 = (ts * 12) / 10; /* output.c:218 */
    int n = 0;

    if (outformat == 0 || outformat == 1) {
        if (outformat == 0) {
            pkt[n++] = '*'; /* output.c:223 */
        } else {
            n += sprintf(pkt, "@%012llX", (unsigned long long)(ts12 & 0xffffffffffffULL));
        }
        for (int i = 0; i < len; i++) { /* "%02X", output.c:247-249 */
            pkt[n++] = hex[frame[i] >> 4];
            pkt[n++] = hex[frame[i] & 15];
        }
        pkt[n++] = ';';
        pkt[n++] = '\n';
        pkt[n] = 0;
        return n;
    }
    /* Beast, output.c:230-243,253-259.  lvl is a uint8_t assigned from a double
     * expression: nearbyint(sqrt(pw))/8/5, truncated by the conversion. */
    uint8_t lvl = (uint8_t)(nearbyint(sqrt((double)pw)) / 8 / 5);
    pkt[n++] = 0x1a;
    pkt[n++] = (len == 7) ? '2' : '3';
    for (int sh = 40; sh >= 0; sh -= 8) {
        char ch = (char)(ts12 >> sh);
        pkt[n++] = ch;
        if (ch == 0x1a)
            pkt[n++] = ch;
    }
    pkt[n++] = (char)lvl; /* not escaped */
    for (int i = 0; i < len; i++) {
        char ch = (char)frame[i];
        pkt[n++] = ch;
        if (ch == 0x1a)
            pkt[n++] = ch;
    }
    return n;
}
