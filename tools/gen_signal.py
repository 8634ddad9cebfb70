"""gen_signal.py -- the build's own synthetic-capture generator (bench workloads, tests).

Synthetic Airspy-R2-style captures for the adsbdec "-f" path:
real uint16 samples at 20 MS/s carrying a 12-bit ADC code centred on 2048
(air.c:64 `(float)r[i]-0x800`), a carrier at fs/4, Mode-S PPM envelopes and
Gaussian noise.  This is the build's own generator; the reference has none.

    x[n] = clip(rint(2048 + E[n]*cos(pi*n/2 + phi) + N(0, sigma)), 0, 4095)

Envelope of one frame starting at sample s (20 samples per microsecond):
  preamble pulses of 10 samples at s+0, s+20, s+70, s+90;
  data bit i at s+160+20*i: first 10 samples high for 1, last 10 high for 0.
"""
from __future__ import annotations

import numpy as np

POLY = 0xFFF409  # crc.h generator (Mode-S CRC-24)
FRAME_SAMPLES_LONG = 160 + 20 * 112   # 2400 input samples
FRAME_SAMPLES_SHORT = 160 + 20 * 56   # 1280


def _crc_table():
    t = []
    for b in range(256):
        c = b << 16
        for _ in range(8):
            c = ((c << 1) ^ POLY) if (c & 0x800000) else (c << 1)
            c &= 0xFFFFFF
        t.append(c)
    return t


_CRC_TABLE = _crc_table()


def crc24(data: bytes) -> int:
    """MSB-first CRC-24, zero init (crc.h:36-38 semantics); the table is generated from the polynomial above."""
    c = 0
    for b in data:
        c = ((c << 8) & 0xFFFFFF) ^ _CRC_TABLE[b ^ (c >> 16)]
    return c


def make_frame(df: int, rng: np.random.Generator, payload: bytes | None = None) -> bytes:
    """A frame whose residual is zero (valid.c:51,73): DF11 -> 7 bytes, DF17/18 -> 14."""
    n = 7 if df == 11 else 14
    if payload is None:
        payload = bytes(rng.integers(0, 256, size=n - 4, dtype=np.uint8).tolist())
    head = bytes([(df << 3) | int(rng.integers(0, 8))]) + payload
    c = crc24(head)
    return head + bytes([(c >> 16) & 255, (c >> 8) & 255, c & 255])


def frame_envelope(frame: bytes) -> np.ndarray:
    """0/1 envelope of one frame at 20 samples/us."""
    nbits = 8 * len(frame)
    env = np.zeros(160 + 20 * nbits, dtype=np.float32)
    for s in (0, 20, 70, 90):
        env[s:s + 10] = 1.0
    bits = np.unpackbits(np.frombuffer(frame, dtype=np.uint8))
    first = 160 + 20 * np.arange(nbits) + np.where(bits != 0, 0, 10)     # a 1 is high in the first half of its microsecond
    env[(first[:, None] + np.arange(10)[None, :]).reshape(-1)] = 1.0
    return env


def synth(n_samples: int, frames: list[tuple[int, bytes, float, float]], sigma: float,
          seed: int, dc: float = 2048.0) -> np.ndarray:
    """frames: list of (start_sample, frame_bytes, amplitude, phase)."""
    rng = np.random.default_rng(seed ^ 0x5EED)
    sig = np.zeros(n_samples, dtype=np.float32)
    for start, fr, amp, phi in frames:
        env = frame_envelope(fr)
        end = min(n_samples, start + len(env))
        if end <= start:
            continue
        n = np.arange(start, end)
        sig[start:end] += (amp * env[: end - start] * np.cos(np.pi * n / 2 + phi)).astype(np.float32)
    if sigma > 0:
        sig += rng.normal(0.0, sigma, size=n_samples).astype(np.float32)
    return np.clip(np.rint(sig + dc), 0, 4095).astype(np.uint16)


def sparse_capture(n_samples: int, n_frames: int, seed: int, sigma: float = 8.0,
                   amp=(200.0, 1500.0), dfs=(17,), min_gap: int = 5200):
    """Config-1/2 style input: frames at least `min_gap` samples apart.
    Returns (x, truth) where truth is a list of (start_sample, frame_bytes)."""
    rng = np.random.default_rng(seed)
    slots = n_samples // min_gap
    n_frames = min(n_frames, max(slots - 1, 0))
    chosen = np.sort(rng.choice(slots - 1, size=n_frames, replace=False)) if n_frames else []
    frames, truth = [], []
    for s in chosen:
        start = int(s) * min_gap + int(rng.integers(0, min_gap - 2400))
        df = int(dfs[int(rng.integers(0, len(dfs)))])
        fr = make_frame(df, rng)
        a = float(rng.uniform(*amp))
        phi = float(rng.uniform(0, 2 * np.pi))
        frames.append((start, fr, a, phi))
        truth.append((start, fr))
    return synth(n_samples, frames, sigma, seed), truth


def dense_capture(n_samples: int, seed: int, sigma: float = 300.0, n_frames: int = 0,
                  amp=(1200.0, 2000.0), dfs=(17, 18, 11)):
    """Config-3 style input: wide-band noise (about 7% of offsets pass the preamble
    test) with optional, possibly overlapping, frames on top."""
    rng = np.random.default_rng(seed)
    frames, truth = [], []
    for _ in range(n_frames):
        start = int(rng.integers(0, max(1, n_samples - 2400)))
        df = int(dfs[int(rng.integers(0, len(dfs)))])
        fr = make_frame(df, rng)
        frames.append((start, fr, float(rng.uniform(*amp)), float(rng.uniform(0, 2 * np.pi))))
        truth.append((start, fr))
    return synth(n_samples, frames, sigma, seed), truth


# ---------------------------------------------------------------------------------------------------------------------
# Device-side generators of the BASELINE workloads (bench.py and the -m gpu tests build their captures with these; `torch`
# is passed in so that this module stays importable without it).  tests/test_generators.py pins each of them by digest: an
# edit here cannot silently change what the 256 Mi-sample tests and the benchmark check.
# ---------------------------------------------------------------------------------------------------------------------
FRAME_GAP = 20_000     # one frame slot per millisecond of signal at 20 MS/s
GEN_BLOCK = 32 << 20   # noise is generated in blocks seeded by (seed, block): any slice of a stream is reproducible


def _frame_plan(n_samples: int, n_frames: int | None, seed: int, df11_share: float, damage_share: float, amp):
    """Which frames go where; every per-frame detail comes from a generator of its own slot,
    so that a slice of the stream can be synthesised without the rest."""
    rng = np.random.default_rng(seed)
    slots = n_samples // FRAME_GAP
    if n_frames is None:
        n_frames = slots
    n_frames = min(n_frames, slots)
    which = np.sort(rng.choice(slots, size=n_frames, replace=False)) if n_frames else np.empty(0, int)
    starts = which * FRAME_GAP + rng.integers(0, FRAME_GAP - 2400, size=n_frames)
    is11 = rng.random(n_frames) < df11_share
    damaged = rng.random(n_frames) < damage_share if damage_share else np.zeros(n_frames, bool)
    amps = rng.uniform(amp[0], amp[1], n_frames)
    phis = rng.uniform(0, 2 * np.pi, n_frames)
    return which, starts, is11, damaged, amps, phis


def make_workload(torch, n_samples: int, n_frames: int | None = None, seed: int = 1,
                  sigma: float = 8.0, df11_share: float = 0.15, device=None, damage_share: float = 0.0,
                  lo: int = 0, hi: int | None = None, amp=(200.0, 1500.0)):
    """Synthetic capture built ON THE DEVICE: uint16 codes in [0,4095]
    around 2048, fs/4 carrier, PPM frames with valid CRC in ~1 ms slots, Gaussian noise.
    [lo, hi): only that slice of the n_samples-long stream is built (shard mode); the
    result does not depend on how the stream is sliced.
    Returns (int16 cuda tensor viewed as the uint16 stream, truth [(start_sample, frame)])."""
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    hi = n_samples if hi is None else hi
    which, starts, is11, damaged, amps, phis = _frame_plan(n_samples, n_frames, seed, df11_share, damage_share, amp)
    sel_all = np.nonzero((starts + 2400 > lo) & (starts < hi))[0]
    waves = np.zeros((sel_all.size, 2400), dtype=np.float32)
    truth = []
    k = np.arange(2400)
    for row, i in enumerate(sel_all):
        frng = np.random.default_rng([seed, int(which[i])])
        fr = make_frame(11 if is11[i] else 17, frng)
        if damaged[i]:  # one flipped bit: a CRC reject (or a 1-bit repair, extension)
            kbit = int(frng.integers(0, 8 * len(fr)))
            fr = bytes(b ^ ((0x80 >> (kbit & 7)) if j == kbit >> 3 else 0) for j, b in enumerate(fr))
        env = frame_envelope(fr)
        s = int(starts[i])
        waves[row, : env.size] = amps[i] * env * np.cos(np.pi * (s + k[: env.size]) / 2 + phis[i])
        truth.append((s, fr))
    st = starts[sel_all]

    gen = torch.Generator(device=dev)
    out = torch.empty(hi - lo, dtype=torch.int16, device=dev)
    for b in range(lo // GEN_BLOCK, (hi + GEN_BLOCK - 1) // GEN_BLOCK):
        b0, b1 = b * GEN_BLOCK, min(n_samples, (b + 1) * GEN_BLOCK)
        gen.manual_seed(seed * 1_000_003 + b)
        buf = torch.randn(b1 - b0, generator=gen, device=dev, dtype=torch.float32) * sigma
        inside = np.nonzero((st >= b0) & (st + 2400 <= b1))[0]
        straddle = np.nonzero(((st < b1) & (st + 2400 > b1)) | ((st < b0) & (st + 2400 > b0)))[0]
        if inside.size:
            idx = torch.from_numpy((st[inside, None] - b0 + k[None, :]).reshape(-1)).to(dev)
            buf.index_add_(0, idx, torch.from_numpy(waves[inside].reshape(-1)).to(dev))
        for j in straddle:  # frames cut by a generation block boundary
            a, e = max(b0, st[j]), min(b1, st[j] + 2400)
            buf[a - b0: e - b0] += torch.from_numpy(waves[j, a - st[j]: e - st[j]]).to(dev)
        a, e = max(b0, lo), min(b1, hi)
        out[a - lo: e - lo] = torch.clamp(torch.round(buf[a - b0: e - b0] + 2048.0), 0, 4095).to(torch.int16)
        del buf
    return out, truth


def make_dense(torch, n: int, seed: int):
    """BASELINE configs[2]: wide-band noise (sigma = 300 around 2048: ~7 % of all offsets pass the
    preamble test, ~0.65 % the DF gate with -a) with strong 112-bit (DF17) frames on top, one per ms."""
    return make_workload(torch, n, seed=seed, sigma=300.0, df11_share=0.0, amp=(1200.0, 2000.0))[0]


TILE_BLOCK = 31_200_000   # a multiple of 2400 (a 112-bit frame), 20 000 (a 1 ms slot), 260 (a frame start) and 4 (the fs/4 carrier)


def _frame_start_wave(amp=900.0, phi=0.7):
    """Preamble + the first five bits of a DF17 frame (10001), 260 samples: repeated back to back it makes ~30 % of the offsets
    pass the preamble test and 7 % pass the DF gate too (measured with the oracle) -- no CRC ever matches."""
    env = np.zeros(260, np.float32)
    for s0 in (0, 20, 70, 90):
        env[s0:s0 + 10] = 1.0
    for i, b in enumerate((1, 0, 0, 0, 1)):
        s0 = 160 + 20 * i + (0 if b else 10)
        env[s0:s0 + 10] = 1.0
    return (amp * env * np.cos(np.pi * np.arange(260) / 2 + phi)).astype(np.float32)


def make_tiled(torch, n: int, seed: int, sigma: float, storm_share: float, frames: bool, device=None):
    """Captures that are FULL of signal, built on the device block by block: 112-bit DF17 frames packed back to back (256
    distinct ones with valid CRC, drawn at random, amplitude 600-1900) and / or 1 ms slots of frame starts repeated back to
    back (storm_share of the slots), in Gaussian noise."""
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    rng = np.random.default_rng(seed)
    k = np.arange(2400)
    waves = np.zeros((256, 2400), np.float32)
    if frames:
        for r in range(256):
            env = frame_envelope(make_frame(17, rng))
            waves[r, : env.size] = rng.uniform(600, 1900) * env * np.cos(np.pi * k[: env.size] / 2 + rng.uniform(0, 2 * np.pi))
    waves_d = torch.from_numpy(waves).to(dev)
    storm = torch.from_numpy(np.tile(_frame_start_wave(), 77)[:20000].copy()).to(dev)
    gen = torch.Generator(device=dev)
    out = torch.empty(n, dtype=torch.int16, device=dev)
    for b in range((n + TILE_BLOCK - 1) // TILE_BLOCK):
        gen.manual_seed(seed * 1_000_003 + b)
        idx = torch.randint(0, 256, (TILE_BLOCK // 2400,), generator=gen, device=dev)
        sig = waves_d[idx].reshape(-1)
        if storm_share > 0:
            pick = torch.rand(TILE_BLOCK // 20000, generator=gen, device=dev) < storm_share
            sig.view(-1, 20000)[pick] = storm
        sig += torch.randn(TILE_BLOCK, generator=gen, device=dev, dtype=torch.float32) * sigma
        b0, b1 = b * TILE_BLOCK, min(n, (b + 1) * TILE_BLOCK)
        out[b0:b1] = torch.clamp(torch.round(sig[: b1 - b0] + 2048.0), 0, 4095).to(torch.int16)
        del sig, idx
    return out


def make_dense10(torch, n: int, seed: int):
    """BASELINE configs[2] at its stated density: ~10 % of the offsets pass the preamble test (sigma = 300 noise alone gives
    7 %; 112-bit frames packed back to back in that noise 9.4 %; 3 % of the 1 ms slots hold frame starts packed back to back,
    of which 30 % pass).  ~100 k frames decode per 256 Mi samples: eight times the sparse workload."""
    return make_tiled(torch, n, seed, 300.0, 0.03, True)


def make_gate_storm(torch, n: int, seed: int):
    """The adversarial capture: nothing but frame starts (preamble + DF17's five bits) packed back to back, sigma = 30: 30 %
    of the offsets pass the preamble test and 7 % pass the DF gate -- ~3 600 survivors per 48 k-offset tile against a queue
    of 1 024, so EVERY tile falls back to its overflow rounds -- and no CRC ever matches."""
    return make_tiled(torch, n, seed, 30.0, 1.0, False)
