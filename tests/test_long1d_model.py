"""CPU models of the chunked multi-level 1-D kernels (mifwt_dwt1_long.hip, kernel ids 17 / 18).

The launch geometry comes from the library itself (``mifwt_dwt1_long_plan``, host arithmetic); the per-workgroup range bookkeeping of
the kernels is restated here in Python and checked for what the GPU tests cannot see directly: every coefficient of every level is
owned by exactly one workgroup, every read stays inside what the workgroup holds, nothing exceeds the LDS buffers — and an
emulation that computes each chunk ONLY from the data the kernel would hold reproduces the fp64 oracle."""
import ctypes

import numpy as np
import pytest

from oracle import fwt_oracle as O
from ptwt_amd import _engine
from ptwt_amd._wavelets import host_taps

MODES = {"zero": 0, "constant": 1, "reflect": 2, "periodic": 3, "symmetric": 4}


def plan(inverse, flen, mode, rows, n, nlevels, m=None):
    lib = _engine.load_library()
    lib.mifwt_dwt1_long_plan.restype = ctypes.c_int
    lib.mifwt_dwt1_long_plan.argtypes = [ctypes.c_int] * 4 + [ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                                         ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]
    out = (ctypes.c_int32 * 6)()
    marr = (ctypes.c_int32 * len(m))(*m) if m is not None else None
    ok = lib.mifwt_dwt1_long_plan(inverse, 0, flen, mode, rows, n, nlevels, marr, out)
    return list(out) if ok else None


def ext(i, n, mode):
    if 0 <= i < n:
        return i
    if mode == "zero":
        return -1
    if mode == "constant":
        return 0 if i < 0 else n - 1
    if mode == "periodic":
        return i % n
    if mode == "symmetric":
        return -i - 1 if i < 0 else 2 * n - 1 - i
    return -i if i < 0 else 2 * (n - 1) - i


def pieces_of(K, chunk, nchunks, end_l, end_r, nK):
    out = [[(0, end_l), (nK - end_r, nK)]]  # the workgroup that owns both ends of a row
    for c in range(nchunks):
        A = end_l + c * chunk
        out.append([(A, min(A + chunk, nK - end_r))])
    return out


def ranges(A, B, last, K, HL, n, ends):
    """computed [a, b) and owned [oa, ob) per level (index = level), the kernel's closed forms (LongPiece)."""
    a, b, oa, ob = [0] * (K + 1), [0] * (K + 1), [0] * (K + 1), [0] * (K + 1)
    for l in range(K + 1):
        sh = K - l
        va = (A << sh) - HL * ((1 << sh) - 1)
        a[l] = max(0, va) if ends else va
        b[l] = min(n[l], B << sh) if ends else (B << sh)
        oa[l] = A << sh
        ob[l] = (n[l] if last else min(n[l], B << sh)) if ends else (B << sh)
    return a, b, oa, ob


@pytest.mark.parametrize("wavelet,mode,n0,rows,want", [("db5", "periodic", 1000000, 32, 10), ("db4", "reflect", 100003, 3, 8), ("haar", "zero", 65536, 2, 6),
                                                        ("sym10", "symmetric", 40001, 1, 5), ("db2", "constant", 33001, 2, 4), ("db5", "periodic", 15633, 32, 4),
                                                        ("db3", "reflect", 5000, 3, 9), ("db4", "reflect", 4096, 4096, 6), ("db4", "symmetric", 16384, 1024, 8),
                                                        ("db2", "periodic", 6001, 700, 7)])
def test_analysis_chunks_cover_every_level_once_and_fit(wavelet, mode, n0, rows, want):
    lo, hi = host_taps(wavelet)[:2]
    L, HL = len(lo), len(lo) - 2
    p = plan(0, L, MODES[mode], rows, n0, want)
    assert p is not None
    K, chunk, nchunks, end_l, end_r, cap = p
    n = [n0]
    for _ in range(K):
        n.append((n[-1] + L - 1) // 2)
    direct_a, direct_b = cap // 4 + 64, cap // 2 + 64  # LDS floats of buffers A / B when level 1 reads global memory
    owned = [np.zeros(v, dtype=np.int32) for v in n]
    for wi, pcs in enumerate(pieces_of(K, chunk, nchunks, end_l, end_r, n[K])):
        ends = wi == 0
        held = []
        for (A, B) in pcs:
            assert B - A >= (L if ends else 1)
            a, b, oa, ob = ranges(A, B, B == n[K], K, HL, n, ends)
            for l in range(K + 1):
                assert 0 <= a[l] <= oa[l] and ob[l] <= b[l] <= n[l], (wi, l)
                if l >= 1:
                    owned[l][oa[l]:ob[l]] += 1
            held.append((a, b))
        for l in range(1, K):  # what the workgroup parks at level l fits the buffer it goes to
            tot = sum(b[l] - a[l] for a, b in held) + 8 * len(held)
            assert tot + 16 <= (direct_b if l % 2 else direct_a), (wi, l, tot)
        # every sample an output reads is either inside the piece or (end pieces) mapped by the boundary rule into one of the two pieces
        for (a, b), (A, B) in zip(held, pcs):
            for l in range(K):
                for k in (a[l + 1], b[l + 1] - 1):
                    for t in (0, L - 1):
                        q = ext(2 * k - HL + t, n[l], mode)
                        assert q < 0 or any(aa[l] <= q < bb[l] for aa, bb in held), (wi, l, k, q)
                if not ends:
                    assert 2 * a[l + 1] - HL == a[l] and 2 * b[l + 1] == b[l]  # map-free interior: the window arithmetic of the fast body
    for l in range(1, K + 1):
        assert (owned[l] == 1).all(), (l, np.flatnonzero(owned[l] != 1)[:8])


@pytest.mark.parametrize("wavelet,n0,rows,level", [("db5", 1000000, 32, 10), ("db4", 100003, 3, 8), ("haar", 65536, 2, 6), ("sym10", 40001, 1, 5), ("db5", 7821, 32, 3), ("db4", 4096, 4096, 6), ("db2", 1024, 16384, 5),
                                                   ("db4", 16384, 1024, 8)])
def test_synthesis_chunks_cover_the_output_once_and_fit(wavelet, n0, rows, level):
    L = len(host_taps(wavelet)[0])
    HLn = L // 2
    lens = [n0]
    for _ in range(level):
        lens.append((lens[-1] + L - 1) // 2)
    lens = lens[::-1]  # coefficients entering each step, coarsest first; then the output
    out = [2 * lens[s] - L + 2 - ((2 * lens[s] - L + 2) - lens[s + 1]) for s in range(level)]
    assert out == lens[1:]
    k = min(level, 8)
    p = None
    while k >= 2 and p is None:
        p = plan(1, L, 0, rows, 0, k, lens[level - k:])
        k -= 1 if p is None else 0
    assert p is not None
    K, chunk, nchunks, _, _, cap = p
    m = lens[level - K:]
    assert chunk % 4 == 0 and nchunks == -(-m[K] // chunk)
    covered = np.zeros(m[K], dtype=np.int32)
    for c in range(nchunks):
        x0, x1 = c * chunk, min(m[K], (c + 1) * chunk)
        covered[x0:x1] += 1
        lo, hi = x0, x1
        for s in range(K - 1, -1, -1):
            hi = min(m[s], ((hi - 1) >> 1) + HLn)
            lo >>= 1
            assert 0 <= lo < hi <= m[s]
            big = (K - 1 - s) % 2 == 0  # step s reads buffer A when K - 1 - s is even
            assert hi - lo + HLn + 8 <= (cap if big else cap // 2 + 64), (c, s)
    assert (covered == 1).all()


@pytest.mark.parametrize("mode", ["periodic", "reflect", "zero", "constant", "symmetric"])
def test_chunked_analysis_emulation_matches_oracle(mode):
    """Each workgroup's outputs computed ONLY from the samples its pieces hold (end pieces through the boundary map, possibly into the
    other piece) — the algorithm of dwt1_long_body — against the oracle's whole-row transform."""
    wavelet, n0, want = "db3", 40960 + 37, 5
    lo, hi = (np.array(t) for t in host_taps(wavelet)[:2])
    L, HL = len(lo), len(lo) - 2
    K, chunk, nchunks, end_l, end_r, cap = plan(0, L, MODES[mode], 2, n0, want)
    rng = np.random.default_rng(5)
    x = rng.standard_normal(n0)
    n = [n0]
    for _ in range(K):
        n.append((n[-1] + L - 1) // 2)
    det = [np.full(v, np.nan) for v in n]
    approx = np.full(n[K], np.nan)
    for wi, pcs in enumerate(pieces_of(K, chunk, nchunks, end_l, end_r, n[K])):
        ends = wi == 0
        rg = [ranges(A, B, B == n[K], K, HL, n, ends) for (A, B) in pcs]
        held = [{q: x[q] for q in range(a[0], b[0])} for (a, b, _, _) in rg]  # level-0 samples per piece
        for l in range(K):
            nxt = []
            for pi, (a, b, oa, ob) in enumerate(rg):
                cur = {}
                for k in range(a[l + 1], b[l + 1]):
                    acc_lo = acc_hi = 0.0
                    for t in range(L):
                        q = ext(2 * k - HL + t, n[l], mode)
                        if q < 0:
                            continue
                        v = held[pi][q] if q in held[pi] else held[1 - pi][q]  # (KeyError = the kernel would read what it does not hold)
                        acc_lo += lo[L - 1 - t] * v
                        acc_hi += hi[L - 1 - t] * v
                    cur[k] = acc_lo
                    if oa[l + 1] <= k < ob[l + 1]:
                        assert np.isnan(det[l + 1][k])
                        det[l + 1][k] = acc_hi
                nxt.append(cur)
            held = nxt
        for (A, B), cur in zip(pcs, held):
            approx[A:B] = [cur[k] for k in range(A, B)]
    want_c = O.wavedec(x[None], wavelet, mode=mode, level=K)
    assert np.allclose(approx, want_c[0][0], rtol=0, atol=1e-12)
    for l in range(1, K + 1):
        assert np.allclose(det[l], want_c[K + 1 - l][0], rtol=0, atol=1e-12), l
