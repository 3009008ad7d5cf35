"""CPU model of the two matrix-core kernels for long filters (csrc/mifwt_dwt2_fwd_mfma.hip, kernel id 11, and
csrc/mifwt_dwt2_inv_mfma.hip, kernel id 23): the banded matrices T / S a filter bank becomes over a block of 16 outputs (the GEMM the
kernels issue as v_mfma_f32_32x32x16_f16), and the WALK down a column panel — which input rows / coefficient rows travel in which chunk,
the two halves of the LDS ring and the order the vertical pass reads them in, the priming chunk of a unit, the segments of a panel, the
window columns that are patched / zeroed at the plane's edges.  Everything in fp64 (the kernels' f16 roundings are what the GPU tests
bound); the result must equal the oracle's level (src/ptwt/conv_transform_2.py:142-149 and :222-249) to rounding."""
import numpy as np
import pytest

from oracle import fwt_oracle as O

MR, MC, IC, WR = 16, 64, 160, 32  # analysis: output rows / columns of a tile, window columns, input rows of a chunk
SR, SC, SOC = 16, 80, 128         # synthesis: coefficient rows / columns of a chunk, output columns of a tile


def window_shift(L):
    """The window of the walk kernel starts L - 2 + s samples before the first output's pair, s = the fewest samples that make that a
    multiple of 8 (16-byte pieces of a row then start on 16-byte boundaries); the 64-sample window still holds the 16 outputs' taps."""
    s = (8 - (L - 2) % 8) % 8
    assert 2 * 15 + L - 1 + s <= 63
    return s


def analysis_T(dec_lo, dec_hi, shift=0):
    """T[(band, k), j] = h_band[2k + L - 1 + shift - j]: 16 low-pass and 16 high-pass outputs from a window of 64 samples that starts
    L - 2 + shift samples before the first output's pair."""
    L = len(dec_lo)
    T = np.zeros((32, 64))
    for band, h in enumerate((dec_lo, dec_hi)):
        for k in range(16):
            for j in range(64):
                m = 2 * k + L - 1 + shift - j
                if 0 <= m < L:
                    T[16 * band + k, j] = h[m]
    return T


def synthesis_S(rec_lo, rec_hi):
    """S[2q + r, 32 b + q + i] = g_b[L - 2 - 2i + r]: 32 output samples (16 pairs) from 32 low-band and 32 high-band coefficients."""
    L = len(rec_lo)
    S = np.zeros((32, 64))
    for b, g in enumerate((rec_lo, rec_hi)):
        for q in range(16):
            for r in range(2):
                for i in range(L // 2):
                    S[2 * q + r, 32 * b + q + i] = g[L - 2 - 2 * i + r]
    return S


@pytest.mark.parametrize("wavelet", ["db9", "db12", "sym16"])
def test_banded_matrices_are_the_filter_bank(wavelet):
    rng = np.random.default_rng(1)
    dec_lo, dec_hi, rec_lo, rec_hi = O.filter_bank(wavelet)
    L = len(dec_lo)
    # analysis: a zero-mode 1-D level of a long row, block by block
    x = rng.standard_normal(400)
    a, d = O.wavedec(x[None], wavelet, mode="zero", level=1)
    T = analysis_T(dec_lo, dec_hi)
    xe = np.concatenate([np.zeros(L - 2), x, np.zeros(200)])  # extended coordinates: position e <-> xe[e + L - 2]
    for k0 in range(0, a.shape[-1] - 16, 16):
        out = T @ xe[2 * k0: 2 * k0 + 64]
        assert np.allclose(out[:16], a[0, k0: k0 + 16], atol=1e-12) and np.allclose(out[16:], d[0, k0: k0 + 16], atol=1e-12)
    assert (T[:, 2 * 15 + L:] == 0).all()  # the last window columns never count (L <= 32)
    # synthesis: blocks of 16 output pairs of the cropped reconstruction
    ca, cd = rng.standard_normal(120), rng.standard_normal(120)
    y = O.waverec([ca[None], cd[None]], wavelet)[0]
    S = synthesis_S(rec_lo, rec_hi)
    pad = np.zeros(40)
    ae, de = np.concatenate([ca, pad]), np.concatenate([cd, pad])
    for p0 in range(0, len(y) // 2 - 16, 16):
        out = S @ np.concatenate([ae[p0: p0 + 32], de[p0: p0 + 32]])
        n = min(32, len(y) - 2 * p0)
        assert np.allclose(out[:n], y[2 * p0: 2 * p0 + n], atol=1e-12)


def units_of(tiles_r, seg_tiles):
    """A panel's row segments: (first tile row, tiles)."""
    return [(t0, min(seg_tiles, tiles_r - t0)) for t0 in range(0, tiles_r, seg_tiles)]


def walk_analysis(x, wavelet, mode, seg_tiles):
    """One level of one image the way kernel 11 walks it."""
    dec_lo, dec_hi, _, _ = O.filter_bank(wavelet)
    L = len(dec_lo)
    H, W = x.shape
    Ho, Wo = (H + L - 1) // 2, (W + L - 1) // 2
    sh = window_shift(L)
    T = analysis_T(dec_lo, dec_hi, sh)
    out = np.full((4, Ho, Wo), np.nan)
    rmap = lambda e: O.ext_index([e], H, mode)[0]  # noqa: E731  (-1 = an implicit zero)
    cmap = lambda e: O.ext_index([e], W, mode)[0]  # noqa: E731
    r_end = 2 * Ho
    for tc in range((Wo + MC - 1) // MC):
        k0 = tc * MC
        c_first = 2 * k0 - (L - 2) - sh
        # the patch ranges of csrc/mifwt_dwt2_fwd_mfma.hip:patch_cols
        nl = min(IC, (max(0, -c_first) + 7) & ~7)
        nr0 = max(nl, min(IC, (W - c_first) & ~1))
        nr1 = max(nr0, min(IC, 2 * (min(k0 + MC, Wo) - k0) + L - 2 + sh))
        for tr0, nt in units_of((Ho + MR - 1) // MR, seg_tiles):
            ring = np.full((2, MC, 64), np.nan)  # [horizontal band][output column][ring row]
            for g in range(nt + 1):  # chunk g; chunk 0 primes the ring
                r_first = 2 * MR * tr0 - (L - 2) - sh + WR * g
                chunk = np.zeros((WR, IC))
                for r in range(WR):
                    ri = r_first + r
                    if ri >= r_end:
                        continue  # (feeds no stored row: requested out of range)
                    sr = rmap(ri)
                    for wc in range(IC):
                        ci = c_first + wc
                        if wc >= nr1:
                            chunk[r, wc] = 0.0  # zeroed: feeds no stored column
                        elif wc < nl or wc >= nr0:
                            sc = cmap(ci)  # patched from its boundary-mapped sample
                            chunk[r, wc] = 0.0 if (sr < 0 or sc < 0) else x[sr, sc]
                        else:
                            assert 0 <= ci < W  # the DMA'd part of the row is inside the plane
                            chunk[r, wc] = 0.0 if sr < 0 else x[sr, ci]
                # horizontal pass: 4 blocks of 16 output columns, rows of the chunk -> ring half g & 1 (transposed)
                for kb in range(4):
                    d = chunk[:, 32 * kb: 32 * kb + 64] @ T.T  # [row][(band, kq)]
                    for band in range(2):
                        ring[band, 16 * kb: 16 * kb + 16, 32 * (g & 1): 32 * (g & 1) + 32] = d[:, 16 * band: 16 * band + 16].T
                if g == 0:
                    continue
                # vertical pass of tile tr0 + g - 1: window rows 0..31 = chunk g - 1, 32..63 = chunk g
                old = 32 * ((g - 1) & 1)
                order = [(k + old) & 63 for k in range(64)]
                j0 = (tr0 + g - 1) * MR
                for bh in range(2):
                    win = ring[bh][:, order]  # [column][window row]
                    assert not np.isnan(win).any()
                    d = win @ T.T  # [column][(vertical band, row)]
                    for bv in range(2):
                        for jr in range(MR):
                            if j0 + jr < Ho:
                                n = min(MC, Wo - k0)
                                out[2 * bv + bh, j0 + jr, k0: k0 + n] = d[:n, 16 * bv + jr]
    assert not np.isnan(out).any()
    return out


@pytest.mark.parametrize("wavelet,shape,mode,seg", [("sym16", (150, 200), "reflect", 3), ("db10", (131, 259), "periodic", 2), ("db9", (96, 97), "zero", 1),
                                                     ("db12", (64, 170), "symmetric", 100), ("sym16", (77, 95), "constant", 2)])
def test_analysis_walk_matches_oracle(wavelet, shape, mode, seg):
    rng = np.random.default_rng(3)
    x = rng.standard_normal(shape)
    want = O.wavedec2(x[None], wavelet, mode=mode, level=1)
    got = walk_analysis(x, wavelet, mode, seg)
    # engine band order: bit 1 = vertical high, bit 0 = horizontal high; the oracle returns (H = 'da', V = 'ad', D = 'dd')
    for s, w in ((0, want[0][0]), (2, want[1][0][0]), (1, want[1][1][0]), (3, want[1][2][0])):
        assert np.abs(got[s] - w).max() < 1e-11 * max(1.0, np.abs(w).max()), s


def walk_synthesis(bands, wavelet, out_hw, seg_tiles):
    """One level of one image the way kernel 23 walks it.  bands = [aa, ad, da, dd] (second letter = along the rows)."""
    _, _, rec_lo, rec_hi = O.filter_bank(wavelet)
    S = synthesis_S(rec_lo, rec_hi)
    Mh, Mw = bands[0].shape
    H, W = out_hw
    y = np.full((H, W), np.nan)
    for tc in range((W + SOC - 1) // SOC):
        c0 = (SOC // 2) * tc
        for tr0, nt in units_of((H + 2 * SR - 1) // (2 * SR), seg_tiles):
            ring = np.full((2, SOC, 32), np.nan)  # [vertical band][output column][ring row]
            for g in range(nt + 1):
                r0 = SR * (tr0 + g)
                chunk = np.zeros((4, SR, SC))
                for s in range(4):
                    for r in range(SR):
                        if r0 + r < Mh:  # rows past the end: requested out of range
                            n = max(0, min(SC, Mw - c0))  # columns past the end: zeroed in LDS
                            chunk[s, r, :n] = bands[s][r0 + r, c0: c0 + n]
                # horizontal pass: rows n = (vertical band, coefficient row), K = (low band | high band) x 32 columns
                for kb in range(4):
                    for bv in range(2):
                        a = np.concatenate([chunk[2 * bv + 0][:, 16 * kb: 16 * kb + 32], chunk[2 * bv + 1][:, 16 * kb: 16 * kb + 32]], axis=1)
                        d = a @ S.T  # [coefficient row][32 output columns]
                        ring[bv, 32 * kb: 32 * kb + 32, 16 * (g & 1): 16 * (g & 1) + 16] = d.T
                if g == 0:
                    continue
                oldh = (g - 1) & 1
                order = [16 * (((k >> 4) + oldh) & 1) + (k & 15) for k in range(32)]
                row0 = 2 * SR * (tr0 + g - 1)
                win = np.concatenate([ring[0][:, order], ring[1][:, order]], axis=1)  # [output column][K]
                assert not np.isnan(win).any()
                d = win @ S.T  # [output column][32 output rows]
                for m in range(32):
                    if row0 + m < H:
                        n = min(SOC, W - SOC * tc)
                        y[row0 + m, SOC * tc: SOC * tc + n] = d[:n, m]
    assert not np.isnan(y).any()
    return y


@pytest.mark.parametrize("wavelet,shape,seg", [("sym16", (150, 200), 2), ("db10", (131, 259), 1), ("db9", (96, 97), 3), ("db14", (300, 140), 100)])
def test_synthesis_walk_matches_oracle(wavelet, shape, seg):
    rng = np.random.default_rng(4)
    c = O.wavedec2(rng.standard_normal((1, *shape)), wavelet, mode="zero", level=1)
    cq = (rng.standard_normal(c[0].shape), tuple(rng.standard_normal(b.shape) for b in c[1]))
    want = O.waverec2(cq, wavelet)[0]
    # oracle order (H = 'da', V = 'ad', D = 'dd') -> engine order aa, ad, da, dd
    bands = [cq[0][0], cq[1][1][0], cq[1][0][0], cq[1][2][0]]
    got = walk_synthesis(bands, wavelet, want.shape, seg)
    assert np.abs(got - want).max() < 1e-11 * max(1.0, np.abs(want).max())
