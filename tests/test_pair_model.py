"""CPU model of the tile version of the two-level analysis kernel (csrc/mifwt_dwt2_fwd_pair.hip): the shifted window of
ACTUAL level-1 rows / columns a tile computes, which level-1 block it owns (writes details for), and level 2's boundary
extension read from that window through the index map at the plane's edges.  Every window access is range-checked and
every output must be written exactly once; the result is compared with the oracle's two-level wavedec2."""
import numpy as np
import pytest

from oracle import fwt_oracle as O


def _ext(i, n, mode):
    return int(O.ext_index(np.asarray([i]), n, mode)[0])


def _model(x, lo, hi, mode, T2R):
    L = len(lo)
    HL, C1 = L - 2, 64
    T2C = (C1 - HL) // 2
    OC1 = 2 * T2C
    R1 = 2 * T2R + HL
    R0, C0 = 2 * R1 + HL, 2 * C1 + HL
    H0, W0 = x.shape
    H1, W1 = (H0 + L - 1) // 2, (W0 + L - 1) // 2
    H2, W2 = (H1 + L - 1) // 2, (W1 + L - 1) // 2
    assert H1 >= R1 and W1 >= 64
    d1 = np.full((3, H1, W1), np.nan)
    o2 = np.full((4, H2, W2), np.nan)
    kc = np.arange(C1)
    for tr in range(-(-H2 // T2R)):
        for tc in range(-(-W2 // T2C)):
            j2_0, k2_0 = tr * T2R, tc * T2C
            s1r = min(max(2 * j2_0 - HL, 0), H1 - R1)
            s1c = min(max(2 * k2_0 - HL, 0), W1 - C1)
            r_first, c_first = 2 * s1r - HL, 2 * s1c - HL
            rmap = np.array([_ext(r_first + r, H0, mode) for r in range(R0)])
            cmap = np.array([_ext(c_first + c, W0, mode) for c in range(C0)])
            xt = np.where((rmap[:, None] >= 0) & (cmap[None, :] >= 0), x[np.maximum(rmap, 0)[:, None], np.maximum(cmap, 0)[None, :]], 0.0)
            hlo = sum(lo[t] * xt[:, (L - 1 - t) + 2 * kc] for t in range(L))  # [R0, 64]
            hhi = sum(hi[t] * xt[:, (L - 1 - t) + 2 * kc] for t in range(L))
            ll = np.zeros((R1, C1))
            for i1 in range(R1):
                rows = [2 * i1 + (L - 1) - t for t in range(L)]
                assert max(rows) < R0
                aa = sum(lo[t] * hlo[rows[t]] for t in range(L))
                da = sum(hi[t] * hlo[rows[t]] for t in range(L))
                ad = sum(lo[t] * hhi[rows[t]] for t in range(L))
                dd = sum(hi[t] * hhi[rows[t]] for t in range(L))
                ll[i1] = aa
                m1r = s1r + i1
                if 2 * j2_0 <= m1r < 2 * j2_0 + 2 * T2R:
                    m1c = s1c + kc
                    own = (m1c >= 2 * k2_0) & (m1c < 2 * k2_0 + OC1)
                    assert np.isnan(d1[0, m1r, m1c[own]]).all()
                    d1[0, m1r, m1c[own]], d1[1, m1r, m1c[own]], d1[2, m1r, m1c[own]] = ad[own], da[own], dd[own]
            # level 2, horizontal
            hl = np.zeros((R1, T2C, 2))
            for kk in range(T2C):
                k2 = k2_0 + kk
                if k2 >= W2:
                    continue
                for p in range(L):
                    m = _ext(2 * k2 - HL + p, W1, mode)
                    if m < 0:
                        continue
                    idx = m - s1c
                    assert 0 <= idx < C1, (idx, tc, kk, p)
                    hl[:, kk, 0] += lo[L - 1 - p] * ll[:, idx]
                    hl[:, kk, 1] += hi[L - 1 - p] * ll[:, idx]
            # level 2, vertical
            for j2l in range(T2R):
                j2 = j2_0 + j2l
                if j2 >= H2:
                    continue
                acc = np.zeros((4, T2C))
                for t_ in range(L):
                    e = _ext(2 * j2 + 1 - t_, H1, mode)
                    if e < 0:
                        continue
                    loc = e - s1r
                    assert 0 <= loc < R1, (loc, tr, j2l, t_)
                    acc += np.stack([lo[t_] * hl[loc, :, 0], lo[t_] * hl[loc, :, 1], hi[t_] * hl[loc, :, 0], hi[t_] * hl[loc, :, 1]])
                live = k2_0 + np.arange(T2C) < W2
                assert np.isnan(o2[0, j2, k2_0 + np.arange(T2C)[live]]).all()
                o2[:, j2, k2_0 + np.arange(T2C)[live]] = acc[:, live]
    assert not np.isnan(d1).any() and not np.isnan(o2).any()
    return d1, o2


@pytest.mark.parametrize("wavelet,shape,rows", [("db4", (150, 140), 8), ("db4", (151, 139), 4), ("db2", (130, 171), 6), ("haar", (128, 130), 4),
                                                ("db3", (167, 255), 12), ("db4", (97, 300), 8)])
@pytest.mark.parametrize("mode", ["reflect", "zero", "constant", "symmetric"])
def test_tile_pair_kernel_index_logic(wavelet, shape, rows, mode):
    rng = np.random.default_rng(1)
    fb = O.filter_bank(wavelet)
    x = rng.standard_normal(shape)
    d1, o2 = _model(x, np.asarray(fb[0]), np.asarray(fb[1]), mode, rows)
    want = O.wavedec2(x[None], wavelet, mode=mode, level=2)
    pairs = [(o2[0], want[0][0]), (o2[2], want[1][0][0]), (o2[1], want[1][1][0]), (o2[3], want[1][2][0]),
             (d1[1], want[2][0][0]), (d1[0], want[2][1][0]), (d1[2], want[2][2][0])]
    for got, ref in pairs:
        assert np.abs(got - ref).max() < 1e-12
