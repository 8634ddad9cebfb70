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


def crc24(data: bytes) -> int:
    """Bitwise MSB-first CRC-24, zero init (crc.h:36-38 semantics)."""
    c = 0
    for b in data:
        c ^= b << 16
        for _ in range(8):
            c = ((c << 1) ^ POLY) if (c & 0x800000) else (c << 1)
            c &= 0xFFFFFF
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
    for i, b in enumerate(bits):
        s = 160 + 20 * i + (0 if b else 10)
        env[s:s + 10] = 1.0
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
