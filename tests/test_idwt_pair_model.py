"""CPU model of the two-level synthesis kernel's tile geometry (csrc/mifwt_idwt2_pair.hip): which coarser-level
coefficient rows / columns sit under a workgroup's approximation tile (first row even, first column of either parity),
the scratch-tile extents NR2 / NC2, the polyphase formulas with out-of-range coefficients read as zero, the parity offset
of the horizontal pass, and the crop between the levels (the finer level's coefficient extents ARE the cropped output
extents of the coarser one).  Every index into a modelled LDS tile is range-checked; the result is compared with the
oracle's two-level waverec2.  (On the GPU the kernel itself is compared bit-for-bit with the per-level kernels.)"""
import numpy as np
import pytest

from oracle import fwt_oracle as O

TRO = 32


def _model(a2, d2, d1, lo, hi, out_ext):
    """a2, d2 = (ad2, da2, dd2): coarser bands [M2h, M2w]; d1 = (ad1, da1, dd1): finer details [M1h, M1w]."""
    L = len(lo)
    HL = L // 2
    NQ = 64 - (HL - 1)
    CR = TRO // 2 + HL - 1
    NR2 = (CR - 1) // 2 + HL
    NC2 = 32 + HL
    NP2 = (CR + 1) // 2
    M2h, M2w = a2.shape
    M1h, M1w = d1[0].shape
    H, W = out_ext
    tlo = [(lo[2 * j], lo[2 * j + 1]) for j in range(HL)]
    thi = [(hi[2 * j], hi[2 * j + 1]) for j in range(HL)]
    bands2 = [a2, d2[0], d2[1], d2[2]]  # aa, ad, da, dd
    y = np.full((H, W), np.nan)
    for tr in range(-(-H // TRO)):
        for tc in range(-(-W // (2 * NQ))):
            q0, y0 = tc * NQ, tr * TRO
            m0 = y0 >> 1
            assert m0 % 2 == 0
            p0r, p0c = m0 >> 1, q0 >> 1
            # A. tiles (out-of-range coefficients read as zero)
            c2 = np.zeros((4, NR2, NC2))
            for s in range(4):
                for r in range(NR2):
                    for c in range(NC2):
                        if p0r + r < M2h and p0c + c < M2w:
                            c2[s, r, c] = bands2[s][p0r + r, p0c + c]
            ct = np.zeros((4, CR, 64))
            for s in range(3):
                for r in range(CR):
                    for c in range(64):
                        if m0 + r < M1h and q0 + c < M1w:
                            ct[1 + s, r, c] = d1[s][m0 + r, q0 + c]
            # B1. coarser vertical synthesis
            xt2 = np.zeros((2 * NP2, NC2, 2))
            for pp in range(NP2):
                xl = np.zeros((NC2, 2))
                xh = np.zeros((NC2, 2))
                for i in range(HL):
                    assert pp + i < NR2
                    tl, th = tlo[HL - 1 - i], thi[HL - 1 - i]
                    aa, ad, da, dd = (c2[s, pp + i] for s in range(4))
                    xl[:, 0] += tl[0] * aa + th[0] * da
                    xl[:, 1] += tl[1] * aa + th[1] * da
                    xh[:, 0] += tl[0] * ad + th[0] * dd
                    xh[:, 1] += tl[1] * ad + th[1] * dd
                xt2[2 * pp, :, 0], xt2[2 * pp, :, 1] = xl[:, 0], xh[:, 0]
                xt2[2 * pp + 1, :, 0], xt2[2 * pp + 1, :, 1] = xl[:, 1], xh[:, 1]
            # B2. coarser horizontal synthesis -> ct[0]
            written = np.zeros((CR, 64), bool)
            for r in range(CR):
                for cl in range(33):
                    o = np.zeros(2)
                    for t in range(HL):
                        assert cl + t < NC2
                        w = xt2[r, cl + t]
                        o[0] += tlo[HL - 1 - t][0] * w[0] + thi[HL - 1 - t][0] * w[1]
                        o[1] += tlo[HL - 1 - t][1] * w[0] + thi[HL - 1 - t][1] * w[1]
                    xo = 2 * cl - (q0 & 1)
                    for k in range(2):
                        if 0 <= xo + k < 64:
                            ct[0, r, xo + k] = o[k]
                            written[r, xo + k] = True
            assert written.all()
            # C. finer level
            xt = np.zeros((TRO, 64, 2))
            for pp in range(TRO // 2):
                for i in range(HL):
                    tl, th = tlo[HL - 1 - i], thi[HL - 1 - i]
                    aa, ad, da, dd = (ct[s, pp + i] for s in range(4))
                    xt[2 * pp, :, 0] += tl[0] * aa + th[0] * da
                    xt[2 * pp + 1, :, 0] += tl[1] * aa + th[1] * da
                    xt[2 * pp, :, 1] += tl[0] * ad + th[0] * dd
                    xt[2 * pp + 1, :, 1] += tl[1] * ad + th[1] * dd
            for r in range(TRO):
                for lane in range(NQ):
                    x = 2 * (q0 + lane)
                    if y0 + r >= H or x >= W:
                        continue
                    o = np.zeros(2)
                    for i in range(HL):
                        w = xt[r, lane + i]  # lane + i <= NQ - 1 + HL - 1 = 63
                        o[0] += tlo[HL - 1 - i][0] * w[0] + thi[HL - 1 - i][0] * w[1]
                        o[1] += tlo[HL - 1 - i][1] * w[0] + thi[HL - 1 - i][1] * w[1]
                    assert np.isnan(y[y0 + r, x])
                    y[y0 + r, x] = o[0]
                    if x + 1 < W:
                        y[y0 + r, x + 1] = o[1]
    assert not np.isnan(y).any()
    return y


@pytest.mark.parametrize("wavelet,shape", [("db4", (130, 257)), ("db4", (131, 250)), ("db2", (97, 140)), ("haar", (128, 130)), ("db3", (70, 191))])
def test_idwt_pair_tile_geometry(wavelet, shape):
    rng = np.random.default_rng(3)
    x = rng.standard_normal(shape)
    c = O.wavedec2(x[None], wavelet, mode="reflect", level=2)
    want = O.waverec2(c, wavelet)[0]
    fb = O.filter_bank(wavelet)
    lo, hi = np.asarray(fb[2]), np.asarray(fb[3])
    a2 = c[0][0]
    h2, v2, dd2 = (t[0] for t in c[1])  # (H, V, D) = (da, ad, dd)
    h1, v1, dd1 = (t[0] for t in c[2])
    got = _model(a2, (v2, h2, dd2), (v1, h1, dd1), lo, hi, want.shape)
    assert np.abs(got - want).max() < 1e-12
