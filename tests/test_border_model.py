"""CPU model of the border part of a 2-D analysis adjoint as `adjoint_border2_kernel` computes it (csrc/mifwt_adjoint_border.hip,
round 5): the backward of F.pad + F.conv2d(stride 2) (reference src/ptwt/conv_transform_2.py:142-149) = the zero-mode adjoint over the
whole plane (a synthesis launch) + the terms of every pad position folded onto the border samples.  The kernel gives every border
sample ONE owner — a column owns its top and bottom border rows, an interior row its left and right border columns —, synthesises the
coefficient rows within reach of the line's two frames once (a bounded LDS array per line) and folds them through the preimage
ranges.  This file follows that bookkeeping index by index (ownership, the frames' coefficient ranges and their bound, the frame a
preimage is looked up in) and compares the result with the exact transpose of the oracle's analysis level, for every boundary
extension and 2 .. 12 taps, odd and even extents.  The GPU tests compare the kernel itself with the generic adjoint passes."""
import numpy as np
import pytest

from oracle import fwt_oracle as O

MODES = ["reflect", "symmetric", "periodic", "constant"]


def preimages(n, N, pl, pr, mode):
    """The extended indices that map to sample n as up to three ranges [a, b] (the kernel's `preimages`)."""
    a, b = [n, 0, 0], [n, -1, -1]
    if mode == "constant":
        if n == 0:
            a[1], b[1] = -pl, -1
        if n == N - 1:
            a[2], b[2] = N, N + pr - 1
    elif mode == "periodic":
        if n - N >= -pl:
            a[1] = b[1] = n - N
        if n + N < N + pr:
            a[2] = b[2] = n + N
    elif mode == "symmetric":
        if -1 - n >= -pl:
            a[1] = b[1] = -1 - n
        if 2 * N - 1 - n < N + pr:
            a[2] = b[2] = 2 * N - 1 - n
    elif mode == "reflect":
        if n >= 1 and -n >= -pl:
            a[1] = b[1] = -n
        if n <= N - 2 and 2 * (N - 1) - n < N + pr:
            a[2] = b[2] = 2 * (N - 1) - n
    return a, b


def border_model(g, N, lo, hi, mode):
    """g: the four band gradients [4, M0, M1] (band bit 1 = axis 0 high, bit 0 = axis 1 high) -> the border terms [N0, N1] and how
    often each sample was written."""
    L = len(lo)
    M = [g.shape[1], g.shape[2]]
    pl = [L - 2, L - 2]
    pr = [L - 2 + (N[0] & 1), L - 2 + (N[1] & 1)]
    B = [pr[0] + 1, pr[1] + 1]
    krmax = (3 * L) // 2 + 2
    out = np.zeros(N)
    owners = np.zeros(N, int)
    lines = [("col", c) for c in range(N[1])] + [("row", n0) for n0 in range(B[0], N[0] - B[0])]
    for kind, c in lines:
        b, o = (0, 1) if kind == "col" else (1, 0)
        Nb, Mb, Bb, plb, prb, Mo = N[b], M[b], B[b], pl[b], pr[b], M[o]
        # bands by (high along b, high along o)
        gLL, gHH = g[0], g[3]
        gLH, gHL = (g[1], g[2]) if kind == "col" else (g[2], g[1])

        def at(band, kb, ko):
            return band[kb, ko] if kind == "col" else band[ko, kb]

        oa, ob = preimages(c, N[o], pl[o], pr[o], mode) if kind == "col" else ([c, 0, 0], [c, -1, -1])
        for qo in range(3):
            for eo in range(oa[qo], ob[qo] + 1):
                self_o = qo == 0
                ko_lo, ko_hi = max(0, eo >> 1), min(Mo - 1, (eo + L - 2) >> 1)
                X = np.full((2, krmax, 2), np.nan)
                klo = [0, 0]
                for f in range(2):
                    e_first = -plb if f == 0 else (Nb if self_o else Nb - Bb)
                    e_last = (-1 if self_o else Bb - 1) if f == 0 else Nb + prb - 1
                    klo[f] = max(0, e_first >> 1)
                    khi = klo[f] - 1 if e_last < e_first else min(Mb - 1, (e_last + L - 2) >> 1)
                    for kb in range(klo[f], khi + 1):
                        assert kb - klo[f] < krmax, "a frame's coefficient rows exceed the LDS array of a line"
                        xl = xh = 0.0
                        for ko in range(ko_lo, ko_hi + 1):
                            m = 2 * ko + 1 - eo
                            assert 0 <= m < L
                            xl += at(gLL, kb, ko) * lo[m] + at(gLH, kb, ko) * hi[m]
                            xh += at(gHL, kb, ko) * lo[m] + at(gHH, kb, ko) * hi[m]
                        X[f, kb - klo[f]] = (xl, xh)
                for di in range(2 * Bb):
                    t = di if di < Bb else Nb - 2 * Bb + di
                    ta, tb = preimages(t, Nb, plb, prb, mode)
                    s, any_ = 0.0, False
                    for q in range(1 if self_o else 0, 3):
                        for eb in range(ta[q], tb[q] + 1):
                            f = 0 if eb < Bb else 1
                            for kb in range(max(0, eb >> 1), min(Mb - 1, (eb + L - 2) >> 1) + 1):
                                m = 2 * kb + 1 - eb
                                x = X[f, kb - klo[f]]
                                assert not np.isnan(x[0]), "a preimage reaches a coefficient row its frame did not synthesise"
                                s += lo[m] * x[0] + hi[m] * x[1]
                            any_ = True
                    if any_:
                        idx = (t, c) if kind == "col" else (c, t)
                        out[idx] += s
                        if qo == 0:
                            owners[idx] += 1
    return out, owners


def exact_adjoint(g, N, lo, hi, mode):
    """A^T g for the oracle's analysis level A (dense: one level applied to every unit impulse)."""
    gx = np.zeros(N)
    for n0 in range(N[0]):
        for n1 in range(N[1]):
            e = np.zeros(N)
            e[n0, n1] = 1.0
            a0, d0 = O.dwt_axis(e, lo, hi, mode, axis=0)
            aa, ad = O.dwt_axis(a0, lo, hi, mode, axis=1)
            da, dd = O.dwt_axis(d0, lo, hi, mode, axis=1)
            gx[n0, n1] = (aa * g[0]).sum() + (ad * g[1]).sum() + (da * g[2]).sum() + (dd * g[3]).sum()
    return gx


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("wavelet,N", [("haar", (7, 8)), ("db2", (11, 10)), ("db4", (18, 19)), ("db4", (21, 18)), ("db5", (22, 23)), ("db6", (27, 26))])
def test_border_lines_model(mode, wavelet, N):
    lo, hi = (np.asarray(v, dtype=np.float64) for v in O.filter_bank(wavelet)[:2])
    L = len(lo)
    assert min(N) >= 2 * (L + 1), "the fast route's envelope (two disjoint borders, single fold)"
    rng = np.random.default_rng(L * 100 + N[0])
    M = [(n + L - 1) // 2 for n in N]
    g = rng.standard_normal((4, *M))
    border, owners = border_model(g, N, lo, hi, mode)
    zero = exact_adjoint(g, N, lo, hi, "zero")  # what the synthesis launch writes: the sample's own term
    want = exact_adjoint(g, N, lo, hi, mode)
    np.testing.assert_allclose(zero + border, want, rtol=1e-12, atol=1e-12)
    # one owner per border sample: the row slabs are the column threads', the rest of the column slabs the row threads'
    B = [L - 1 + (N[0] & 1), L - 1 + (N[1] & 1)]
    inner = np.zeros(N, bool)
    inner[B[0] : N[0] - B[0], B[1] : N[1] - B[1]] = True
    assert owners.max() <= 1 and not owners[inner].any()
    assert np.all(border[inner] == 0.0)
