"""CPU restatements of three small pieces of index arithmetic the HIP kernels rely on (csrc/mifwt_stream.h, mifwt_dwt1_tail.hip),
checked exhaustively on ranges the kernels use:
  * FastDiv — workgroup-index decomposition by a launch-time constant with one multiply-high (round-up method);
  * Fold1 — the branch-free single-fold boundary map, against the oracle's general ext_index for indices within one period;
  * the interior / edge split of a fused 1-D level (every output exactly once, interior taps inside the row)."""
import random

import numpy as np
import pytest

from oracle import fwt_oracle as O


def _make_fastdiv(d):
    s = 0
    while (1 << s) < d:
        s += 1
    return (((1 << 32) * ((1 << s) - d)) // d + 1) & 0xFFFFFFFF, s


def _fdiv(n, mul, shift):
    return ((((mul * n) >> 32) + n) & 0xFFFFFFFF) >> shift


def test_fastdiv_matches_integer_division():
    rnd = random.Random(7)
    divisors = list(range(1, 3000)) + [rnd.randrange(1, 1 << 27) for _ in range(5000)]
    for d in divisors:
        mul, shift = _make_fastdiv(d)
        assert mul < (1 << 32)
        for n in [0, 1, d - 1, d, d + 1, 2 * d - 1, 2 * d, (1 << 31) - 1] + [rnd.randrange(0, 1 << 31) for _ in range(8)]:
            assert _fdiv(n, mul, shift) == n // d, (d, n)


def _fold1(i, n, mode):
    kneg = -1 if mode in ("reflect", "symmetric", "zero") else 0
    kpos = -1 if mode == "periodic" else 0
    sym = 1 if mode == "symmetric" else 0
    per = kpos
    ki = (i & kpos) - (i & kneg)
    lo_add = (n & per) - sym
    hi_add = 2 * n - 2 + sym if kneg else ((n - 1) & ~per) - (n & per)
    return lo_add + ki if i < 0 else (hi_add + ki if i >= n else i)


@pytest.mark.parametrize("mode", ["reflect", "symmetric", "constant", "periodic"])
def test_fold1_matches_the_general_map_within_one_period(mode):
    for L in (2, 4, 8, 16, 32):
        for n in range(L, L + 70):  # the tile kernels require n >= L
            idx = np.arange(-(L - 2), n + L - 1)
            want = O.ext_index(idx, n, mode)
            got = np.array([_fold1(int(i), n, mode) for i in idx])
            assert (got == want).all(), (mode, L, n)


def test_fused_1d_level_interior_edge_split():
    for L in (2, 4, 8, 10, 20, 32):
        for n in range(1, 300):
            m = (n + L - 1) >> 1
            k_lo = min((L - 2) >> 1, m)
            k_hi = max(min(n >> 1, m), k_lo)
            seen = [0] * m
            for k in range(k_lo, k_hi):
                assert 2 * k + 1 - (L - 1) >= 0 and 2 * k + 1 < n, (L, n, k)
                seen[k] += 1
            for idx in range(k_lo + (m - k_hi)):
                seen[idx if idx < k_lo else k_hi + (idx - k_lo)] += 1
            assert all(v == 1 for v in seen), (L, n)
