"""The two fused forms of scan_kernel.hip's FIR round the same real numbers as the reference's separate operations
(air.c:64-75: in = (float)r[i] - 0x800; sum += dsfilter[k+o] * fbuff[k]).  Checked here exhaustively on the CPU, without
the kernel: every uint16 sample value against every tap."""
import numpy as np

TAPS = [0.012627, 0.025254, 0.037881, 0.050508, 0.063135, 0.075761, 0.088388]  # air.c:36-45 (the filter is symmetric)


def test_fma_with_the_exact_constant_equals_subtract_then_multiply():
    x = np.arange(65536, dtype=np.float32)  # what the typed load delivers: the converted sample
    for lit in TAPS:
        t = np.float32(lit)
        c = np.float32(2048.0) * t
        assert float(c) == 2048.0 * float(t)  # 2048 t is a binary32 number
        for sign in (1.0, -1.0):  # the fs/4 sign of air.c:79-82
            ref = (np.float32(sign) * t) * (x - np.float32(2048.0)) if sign > 0 else t * (np.float32(2048.0) - x)
            # fma(t, x, -2048 t) = round(t x - 2048 t): the products and the difference are exact in binary64
            # (24 + 16 significant bits), so one conversion to binary32 is the single rounding of the fused operation
            fused = (sign * (float(t) * x.astype(np.float64) - float(c))).astype(np.float32)
            assert np.array_equal(ref, fused), lit


def test_the_doubled_tap_is_an_exact_doubling():
    h0, h1 = np.float32(TAPS[0]), np.float32(TAPS[1])
    assert float(h1) == 2.0 * float(h0)  # (float)0.025254 == 2 x (float)0.012627
    v = np.arange(-2048, 65536 - 2048, dtype=np.float32)  # x - 2048 for every sample value
    m = h0 * v
    assert np.array_equal(h1 * v, np.float32(2.0) * m)  # fl(2 h0 v) = 2 fl(h0 v): no rounding in the doubling
    # so s + fl(2 h0 v) = round(s + 2 m) = fma(m, 2, s) for any s: sampled sums of the magnitude the FIR holds
    rng = np.random.default_rng(7)
    s = (rng.standard_normal(v.size) * 300.0).astype(np.float32)
    sep = s + h1 * v
    fused = (s.astype(np.float64) + 2.0 * m.astype(np.float64)).astype(np.float32)  # exact sum in binary64, one rounding
    assert np.array_equal(sep, fused)
