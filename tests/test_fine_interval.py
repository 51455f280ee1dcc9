"""The arithmetic behind k_sync_finish's fast path (welle.io_amd/csrc/k_sync.hip: fine_decided), restated in numpy and checked
against brute force: (1) the error bound of the reference's ordered float32 summation derived from block sums really bounds it,
(2) whenever the interval test calls the int16 fine-corrector step "decided", it is the step the ordered sums give.
The kernel itself is covered by the parity suites (correctors frame by frame against the oracle, both paths taken)."""
import ctypes
import math

import numpy as np
import pytest

N = 75 * 504
U = 2.0 ** -24
ROWS_PER_BLOCK = 8
libm = ctypes.CDLL("libm.so.6")
libm.atan2f.restype = ctypes.c_float
libm.atan2f.argtypes = [ctypes.c_float, ctypes.c_float]


def seq_sum_f32(x):
    return np.add.accumulate(x.astype(np.float32), dtype=np.float32)[-1]          # strictly sequential float32 accumulation


def block_bound(x):
    """E with |float32 sequential sum - exact sum| <= E, from the sums and magnitude sums of blocks of 8 rows of 504"""
    x = x.astype(np.float64)
    q = 0.0; s = 0.0
    for b0 in range(0, N, ROWS_PER_BLOCK * 504):
        blk = x[b0:b0 + ROWS_PER_BLOCK * 504]
        q += len(blk) * (abs(s) + np.abs(blk).sum())
        s += math.fsum(blk)
    return U * q / (1.0 - (N + 1) * U) * (1 + 2.0 ** -30), s


def fine_from_arg(fine_old, a):
    v = float(fine_old) + 0.1 * float(np.float32(a)) / math.pi * 500
    return int(np.int16(int(v)))                                                  # C truncation toward zero, then int16


def decided(fine_old, re, im):
    ere, sre = block_bound(re); eim, sim = block_bound(im)
    lo = lambda v: np.nextafter(np.float32(v), np.float32(-np.inf)) if float(np.float32(v)) > v else np.float32(v)
    hi = lambda v: np.nextafter(np.float32(v), np.float32(np.inf)) if float(np.float32(v)) < v else np.float32(v)
    xl, xh, yl, yh = lo(sre - ere), hi(sre + ere), lo(sim - eim), hi(sim + eim)
    if not (xl > 0 or yl > 0 or yh < 0):
        return None
    c = [np.float32(libm.atan2f(y, x)) for y in (yl, yh) for x in (xl, xh)]
    a_lo, a_hi = min(c), max(c)
    for _ in range(8):
        a_lo = np.nextafter(a_lo, np.float32(-np.inf)); a_hi = np.nextafter(a_hi, np.float32(np.inf))
    n_lo, n_hi = fine_from_arg(fine_old, a_lo), fine_from_arg(fine_old, a_hi)
    return n_lo if n_lo == n_hi else None


@pytest.mark.parametrize("kind", ["positive", "zero_mean", "mixed_scale", "cancelling", "alternating"])
def test_block_bound_holds(kind):
    rng = np.random.RandomState(hash(kind) % 1000)
    for trial in range(6):
        if kind == "positive":
            x = (0.06 + 0.02 * rng.randn(N)).astype(np.float32)
        elif kind == "zero_mean":
            x = (0.02 * rng.randn(N)).astype(np.float32)
        elif kind == "mixed_scale":
            x = (rng.randn(N) * 10.0 ** rng.uniform(-6, 2, N)).astype(np.float32)
        elif kind == "cancelling":
            x = np.concatenate([np.full(N // 2, 0.37, np.float32), np.full(N - N // 2, -0.37, np.float32)]) + (1e-4 * rng.randn(N)).astype(np.float32)
        else:
            x = (((-1.0) ** np.arange(N)) * (1.0 + 0.3 * rng.rand(N))).astype(np.float32)
        e, s = block_bound(x)
        err = abs(float(seq_sum_f32(x)) - math.fsum(x.astype(np.float64)))
        assert err <= e, (kind, trial, err, e)
        assert e < 3e-3 * np.abs(x.astype(np.float64)).sum()                   # and it is not vacuous


def test_decided_means_equal_to_ordered_sums():
    """angles steered next to the corrector's steps (multiples of pi/50): the interval test may decline, never be wrong"""
    rng = np.random.RandomState(7)
    n_decided = n_declined = 0
    for trial in range(60):
        k = rng.randint(-3, 4)
        target = k * math.pi / 50 + rng.choice([0.0, 1.0]) * rng.uniform(-2e-4, 2e-4) + rng.choice([0.0, 1.0]) * rng.uniform(-0.03, 0.03)
        amp = 10.0 ** rng.uniform(-2.5, -0.5)
        re = (amp * math.cos(target) * (1 + 0.2 * rng.randn(N)) + 0.3 * amp * rng.randn(N)).astype(np.float32)
        im = (amp * math.sin(target) * (1 + 0.2 * rng.randn(N)) + 0.3 * amp * rng.randn(N)).astype(np.float32)
        fine_old = int(rng.randint(-400, 400))
        want = fine_from_arg(fine_old, libm.atan2f(seq_sum_f32(im), seq_sum_f32(re)))
        got = decided(fine_old, re, im)
        if got is None:
            n_declined += 1
        else:
            n_decided += 1
            assert got == want, (trial, target, got, want)
    assert n_decided >= 30 and n_declined >= 3, (n_decided, n_declined)
