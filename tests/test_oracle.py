"""CPU tests of the oracle itself: it must be pinned before it is trusted.

 - CRC table and known answers (crc.h)
 - the restatement against the REAL reference deqframe/valid/formatpkt in oracle/_ref
   (skipped on machines without it) on synthetic POWER buffers that fuzz getdf
 - the restatement against the committed golden fixtures (minted through the real
   chain, oracle/make_golden.py)
 - known-answer / property tests of the front end (air.c:54-92)
The pin of the whole chain on uint16 input, real decodeiq included, is
tests/test_oracle_vs_ref.py.
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_cases, golden_records, load_golden, records

TAPS = np.array([0.012627, 0.025254, 0.037881, 0.050508, 0.063135, 0.075761, 0.088388,
                 0.088388, 0.075761, 0.063135, 0.050508, 0.037881, 0.025254, 0.012627]).astype(np.float32)


def test_crc_known_answers(oracle):
    with open(os.path.join(GOLDEN, "crc_kat.json")) as f:
        kat = json.load(f)
    for v in kat["vectors"]:
        assert oracle.crc_residual(bytes.fromhex(v["frame"])) == int(v["residual"], 16)
    assert [oracle.lib().orc_crc_table(i) for i in range(8)] == [int(h, 16) for h in kat["table_first8"]]
    assert oracle.lib().orc_crc_table(255) == int(kat["table_last"][0], 16)


def test_crc_matches_bitwise_definition(oracle):
    from tools.gen_signal import crc24
    rng = np.random.default_rng(0)
    for n in (7, 14):
        for _ in range(200):
            fr = bytes(rng.integers(0, 256, n, dtype=np.uint8).tolist())
            tail = int.from_bytes(fr[-3:], "big")
            assert oracle.crc_residual(fr) == crc24(fr[:-3]) ^ tail


@pytest.mark.parametrize("name", golden_cases())
def test_oracle_reproduces_golden(oracle, name):
    x, rec = load_golden(name)
    frames, stats = oracle.decode(x, df18=rec["df18"])
    assert records(frames) == golden_records(rec)
    assert stats == rec["stats"]
    for f, g in zip(frames, rec["frames"]):
        assert oracle.formatpkt(f["frame"], f["ts"], f["pw"], 0) == g["avr"].encode()
        assert oracle.formatpkt(f["frame"], f["ts"], f["pw"], 1) == g["mlat"].encode()
        assert oracle.formatpkt(f["frame"], f["ts"], f["pw"], 2) == bytes.fromhex(g["beast"])


needs_ref = pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "_ref", "ref_adsbdec")),
                               reason="oracle/_ref not built (needs /root/reference)")


@needs_ref
@pytest.mark.parametrize("seed,df18", [(1, False), (2, True), (3, True)])
def test_oracle_vs_real_reference_on_signals(oracle, seed, df18):
    from tools import gen_signal as G
    x, _ = G.dense_capture(3 << 17, seed=seed, sigma=30.0 * seed, n_frames=150)
    a = oracle.power(x)
    rf, rstats = oracle.ref_demod(a, df18)
    of, ostats = oracle.decode(x, df18)
    assert [(f["ts"], f["pw"], f["frame"]) for f in of] == [(f["ts"], f["pw"], f["frame"]) for f in rf]
    assert ostats == rstats
    for f, r in zip(of, rf):
        assert oracle.formatpkt(f["frame"], f["ts"], f["pw"], 0) == r["avr"]
        assert oracle.formatpkt(f["frame"], f["ts"], f["pw"], 1) == r["mlat"]
        assert oracle.formatpkt(f["frame"], f["ts"], f["pw"], 2) == r["beast"]


def _plant(a, g, frame, rng):
    """Write an ideal PPM frame into a power array at offset g (10 samples / us)."""
    hi = lambda: float(rng.uniform(4e4, 9e4))
    a[g:g + 80] = rng.uniform(0, 50, 80)
    for s in (0, 10, 35, 45):
        a[g + s:g + s + 5] = hi()
    bits = np.unpackbits(np.frombuffer(frame, dtype=np.uint8))
    for i, b in enumerate(bits):
        l = g + 80 + 10 * i
        a[l:l + 10] = rng.uniform(0, 50, 10)
        a[l + (0 if b else 5):l + (5 if b else 10)] = hi()


@needs_ref
def test_oracle_demod_vs_real_reference_on_random_power(oracle):
    """Fuzzes getdf / getabyte / greedy skip / ts on synthetic POWER (no front end):
    exponential noise makes ~8 % of offsets pass the preamble test and planted
    frames (some overlapping, some straddling a deqframe call boundary) are accepted."""
    from tools import gen_signal as G
    rng = np.random.default_rng(5)
    a = rng.exponential(1000.0, 400_000).astype(np.float32)
    for k in range(150):
        g = int(rng.integers(0, a.size - 1300))
        _plant(a, g, G.make_frame(int(rng.choice([11, 17, 18])), rng), rng)
    for g in (39_700, 39_779, 39_781, 40_979 - 1196, 79_500):   # around T-1200 horizons
        _plant(a, g, G.make_frame(17, rng), rng)
    for df18 in (False, True):
        rf, rstats = oracle.ref_demod(a, df18)
        of, ostats = oracle.demod_power(a, df18)
        assert [(f["ts"], f["pw"], f["frame"]) for f in of] == [(f["ts"], f["pw"], f["frame"]) for f in rf]
        assert ostats == rstats
        assert len(rf) > 60 and rstats["try"][11] > 500


# ------------------------- front end (air.c:54-92) -------------------------
def _expected_power_naive(x):
    """Independent numpy float32 model of air.c:59-92: explicit 14-slot ring,
    summation in physical slot order."""
    n = x.size - x.size % 4
    ring = np.zeros(14, dtype=np.float32)
    t2 = np.concatenate([TAPS, TAPS])
    out = []
    fidx = 0
    for i in range(0, n, 4):
        for half in range(2):
            for k in range(2):
                v = np.float32(x[i + 2 * half + k]) - np.float32(2048)
                ring[fidx % 14] = -v if half else v
                fidx += 1
            o = 14 - fidx % 14
            si = np.float32(0)
            sq = np.float32(0)
            for k in range(0, 14, 2):
                si = np.float32(si + np.float32(t2[k + o] * ring[k]))
                sq = np.float32(sq + np.float32(t2[k + 1 + o] * ring[k + 1]))
            out.append(np.float32(np.float32(si * si) + np.float32(sq * sq)))
    return np.array(out, dtype=np.float32)


def test_front_end_against_independent_numpy_model(oracle):
    rng = np.random.default_rng(3)
    x = rng.integers(0, 4096, 4 * 700, dtype=np.uint16)
    assert np.array_equal(oracle.power(x), _expected_power_naive(x))


def test_front_end_silence_and_impulse(oracle):
    x = np.full(400, 2048, dtype=np.uint16)
    assert not oracle.power(x).any()
    # an impulse on an I sample meets the even-index taps, newest = T[12] (air.c:69-75)
    x[100] = 2048 + 1000          # n = 100: even -> I; n mod 4 == 0 -> sign +
    a = oracle.power(x)
    m0 = 50                        # first output that contains pair 50
    for age in range(7):
        tap = TAPS[12 - 2 * age]
        assert a[m0 + age] == np.float32(tap * np.float32(1000)) ** 2
    assert a[m0 + 7] == 0 and a[m0 - 1] == 0
    # Q sample: odd taps, newest = T[13]
    x[:] = 2048
    x[103] = 2048 - 500           # n = 103: odd -> Q; n mod 4 == 3 -> sign -
    a = oracle.power(x)
    for age in range(7):
        tap = TAPS[13 - 2 * age]
        assert a[51 + age] == np.float32(tap * np.float32(500)) ** 2


def test_front_end_fs4_carrier_gives_flat_power(oracle):
    """A constant-envelope carrier at fs/4 mixes to DC: after the 7-pair warm-up the
    power is constant to rounding, for every carrier phase."""
    n = np.arange(4000)
    for phi in (0.0, 0.7, 2.1):
        x = np.rint(2048 + 1000 * np.cos(np.pi * n / 2 + phi)).astype(np.uint16)
        a = oracle.power(x)[10:]
        assert a.min() > 0 and (a.max() - a.min()) / a.max() < 5e-3


def test_eof_tail_is_never_decoded(oracle):
    """a frame in the last ~41k power samples is not reported."""
    from tools import gen_signal as G
    rng = np.random.default_rng(9)
    fr = G.make_frame(17, rng)
    n = 1 << 18
    early = G.synth(n, [(50_000, fr, 900.0, 0.3)], 5.0, 1)
    late = G.synth(n, [(n - 20_000, fr, 900.0, 0.3)], 5.0, 1)  # g = 121072 > last horizon 119340
    assert [f["frame"] for f in oracle.decode(early)[0]] == [fr]
    assert oracle.decode(late)[0] == []
