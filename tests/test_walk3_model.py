"""CPU model of the depth-walking 3-D kernels' bookkeeping (csrc/mifwt_dwt3_fwd_walk.hip, kernel id 24; csrc/mifwt_dwt3_inv_walk.hip,
kernel id 25): workgroup units (row group x depth segment), the slices a segment walks and the ring slot each one is staged in, the
loader's index maps (rows, slices; zero mode: empty resource -> zeros), the pad columns the compute waves fill in LDS, the column
strips, and the ROLLING depth pass — which accumulator slot an output slice (pair) lives in, which tap meets which slice, when a slot
is re-initialised and when it is complete.  Every LDS slot carries the id of the slice staged in it and every read asserts the id (one
barrier per slice is the kernels' only synchronisation: a request may only overwrite the slot read in the PREVIOUS step).  The
synthesis model stages the bands' contiguous row pieces into NaN-filled slots: a NaN in a stored output would mean that a valid output
read a coefficient behind its piece.  Results are compared with the oracle (reference seams: src/ptwt/conv_transform_3.py:121-141,
205-249).  The GPU tests compare the kernels themselves with the oracle; this file pins the index logic where no GPU is available."""
import numpy as np
import pytest

from oracle import fwt_oracle as O

MODES = ["reflect", "zero", "constant", "periodic", "symmetric"]
KEYS = ["aaa", "aad", "ada", "add", "daa", "dad", "dda", "ddd"]  # band s: bit 2 = depth high, bit 1 = row high, bit 0 = column high


def _fold(i, n, mode):
    """Source index of extended index i, or -1 (zero mode, outside): what Fold1 + the kernels' dead-row tests compute."""
    return int(O.ext_index(np.asarray([i]), n, mode)[0])


# ------------------------------------------------------------------------------------------------------------------ analysis, id 24
def walk3_fwd_model(x, lo, hi, mode, TR, NRG, seg_out, nslots, balanced=True):
    """One batch element [D, H, W] -> the eight bands [Do, Ho, Wo], following dwt3_fwd_walk_kernel unit by unit and step by step."""
    L = len(lo)
    HL, HP = L - 2, L // 2
    D, H, W = x.shape
    Do, Ho, Wo = (D + L - 1) // 2, (H + L - 1) // 2, (W + L - 1) // 2
    IR, IRW = 2 * TR + HL, 2 * TR * NRG + HL
    LP = 4 if L <= 6 else 8
    nstrips = (Wo + 63) // 64
    nq = (Wo + nstrips - 1) // nstrips if balanced else 64
    ngroups = -(-Ho // (TR * NRG))
    nseg = -(-Do // seg_out)
    nrp = 2 * Wo - W
    assert 0 <= nrp <= L - 1 and HL <= LP and nrp <= 8
    out = np.full((8, Do, Ho, Wo), np.nan)
    written = np.zeros((Do, Ho, Wo), int)
    pitch = LP + 512 + 8
    for s in range(nseg):
        zA = s * seg_out
        zB = min(Do, zA + seg_out)
        E0, nsl = 2 * zA - HL, 2 * (zB - zA) + HL
        assert nsl % 2 == 0
        for g in range(ngroups):
            j0 = g * TR * NRG
            r_first = 2 * j0 - HL
            nr_need = 2 * (min(j0 + TR * NRG, Ho) - j0) + HL
            lds = np.full((nslots, IRW, pitch), np.nan)
            lds_id = np.full(nslots, -10**9)
            ib = 0

            def issue(t):
                nonlocal ib
                e = E0 + t
                es = _fold(e, D, mode)
                for i in range(IRW):
                    rs = _fold(r_first + i, H, mode) if i < nr_need else -1
                    row = np.zeros(512)
                    if es >= 0 and rs >= 0:
                        row[:W] = x[es, rs]  # (lanes behind the row: out of the resource's range -> zeros land)
                    lds[ib, i, LP:LP + 512] = row
                lds_id[ib] = t
                ib = (ib + 1) % nslots

            ahead = nslots - 1
            for t in range(min(ahead, nsl)):
                issue(t)
            # compute waves: (strip, row sub-group); accumulators per wave
            acc = {}
            slot = 0
            for t in range(nsl):
                # barrier t: slice t has landed; the loader now requests slice t + ahead WHILE the compute waves read slice t — into the
                # slot slice t - 1 was read from, never the one being read
                assert lds_id[slot] == t, (lds_id[slot], t)
                if t + ahead < nsl:
                    assert ib != slot
                    issue(t + ahead)
                sl = lds[slot]
                # pad columns, by the first / last strip's waves
                for r in range(IRW):
                    for i in range(HL):
                        m = _fold(i - HL, W, mode)
                        sl[r, LP + i - HL] = 0.0 if m < 0 else sl[r, LP + m]
                    for i in range(nrp):
                        m = _fold(W + i, W, mode)
                        sl[r, LP + W + i] = 0.0 if m < 0 else sl[r, LP + m]
                p, PH = t >> 1, t & 1
                R = p % HP
                for sub in range(NRG):
                    jw = j0 + sub * TR
                    for strip in range(nstrips):
                        k0, k1 = strip * nq, min(Wo, (strip + 1) * nq)
                        if k0 >= k1:
                            continue
                        ks = np.arange(k0, k1)
                        rows = sl[2 * sub * TR: 2 * sub * TR + IR]
                        # W pass: samples 2 k - HL + 2 p', + 1 <-> taps L - 1 - 2 p', L - 2 - 2 p'
                        wl = np.zeros((IR, len(ks)))
                        wh = np.zeros((IR, len(ks)))
                        for pp in range(HP):
                            a0 = rows[:, LP + 2 * ks - HL + 2 * pp]
                            a1 = rows[:, LP + 2 * ks - HL + 2 * pp + 1]
                            assert not np.isnan(a0).any() and not np.isnan(a1).any()
                            wl += lo[L - 1 - 2 * pp] * a0 + lo[L - 2 - 2 * pp] * a1
                            wh += hi[L - 1 - 2 * pp] * a0 + hi[L - 2 - 2 * pp] * a1
                        # H pass: output row j of the sub-group <- W-filtered rows 2 j + (L - 1) - m
                        hv = np.zeros((TR, 4, len(ks)))  # components (Ha Wa, Hd Wa, Ha Wd, Hd Wd)
                        for j in range(TR):
                            for m in range(L):
                                r = 2 * j + (L - 1) - m
                                hv[j, 0] += lo[m] * wl[r]
                                hv[j, 1] += hi[m] * wl[r]
                                hv[j, 2] += lo[m] * wh[r]
                                hv[j, 3] += hi[m] * wh[r]
                        # D pass, rolling: output p - q lives in slot (R - q) mod HP; tap m = L - 1 - 2 q - PH
                        A = acc.setdefault((sub, strip), np.zeros((HP, 2, TR, 4, len(ks))))
                        for q in range(HP):
                            sl_q = (R - q + HP) % HP
                            m = L - 1 - 2 * q - PH
                            if q == 0 and PH == 0:
                                A[sl_q] = 0.0  # (a multiply, not an accumulate: the slot's previous output left one pair ago)
                            A[sl_q, 0] += lo[m] * hv
                            A[sl_q, 1] += hi[m] * hv
                        if PH == 1:
                            z = zA + p - (HP - 1)
                            if p >= HP - 1 and z < zB:
                                done = (R + 1) % HP
                                for j in range(TR):
                                    y = jw + j
                                    if y >= Ho:
                                        continue
                                    for dbit in range(2):
                                        for c in range(4):
                                            band = 4 * dbit + 2 * (c & 1) + (c >> 1)  # component c = (H bit = c & 1, W bit = c >> 1)
                                            out[band, z, y, ks] = A[done, dbit, j, c]
                                    written[z, y, ks] += 1
                slot = (slot + 1) % nslots
    assert (written == 1).all(), "every coefficient is stored exactly once"
    return out


FWD_CASES = [
    # (shape, wavelet, TR, NRG, seg_out, nslots, balanced)
    ((21, 19, 37), "db2", 4, 1, 4, 5, True),
    ((12, 13, 141), "db2", 4, 1, 3, 3, True),     # three column strips, ragged row groups
    ((9, 10, 130), "haar", 4, 1, 100, 2, False),  # one segment; strips of 64 + 1
    ((14, 22, 40), "db3", 4, 1, 5, 4, True),
    ((17, 9, 33), "db4", 2, 1, 2, 3, True),       # eight taps: two rows per wave, segments of two output slices
    ((20, 21, 24), "db5", 2, 2, 6, 3, True),      # ten taps: two row sub-groups share a staged slice
    ((18, 17, 70), "db2", 2, 1, 4, 3, True),      # the f64 instances' shapes: two rows per workgroup on small volumes, three staged slices
    ((16, 15, 44), "db3", 2, 1, 5, 3, True),      # f64, six taps: two rows (four would not fit the registers)
]


@pytest.mark.parametrize("case", FWD_CASES)
def test_walk3_fwd_model_vs_oracle(case):
    shape, wavelet, TR, NRG, seg_out, nslots, balanced = case
    rng = np.random.default_rng(len(wavelet) + shape[0])
    bank = O.filter_bank(wavelet)
    lo, hi = [float(v) for v in bank[0]], [float(v) for v in bank[1]]
    x = rng.standard_normal(shape)
    for mode in MODES:
        try:
            want = O.wavedec3(x, wavelet, mode=mode, level=1)
        except RuntimeError:
            continue
        got = walk3_fwd_model(x, lo, hi, mode, TR, NRG, seg_out, nslots, balanced)
        np.testing.assert_allclose(got[0], want[0], rtol=0, atol=1e-12, err_msg=f"{case} {mode} aaa")
        for b in range(1, 8):
            np.testing.assert_allclose(got[b], want[1][KEYS[b]], rtol=0, atol=1e-12, err_msg=f"{case} {mode} {KEYS[b]}")


# ----------------------------------------------------------------------------------------------------------------- synthesis, id 25
def walk3_inv_model(bands, rec_lo, rec_hi, out_shape, CY, seg_out, nslots, st16):
    """bands [8][Md, Mh, Mw] -> the reconstruction [D, H, W] of one batch element, following idwt3_walk_kernel."""
    L = len(rec_lo)
    HL = L // 2
    IY = CY + HL - 1
    Md, Mh, Mw = bands[0].shape
    D, H, W = out_shape
    assert all(n <= 2 * m - L + 2 for n, m in zip(out_shape, (Md, Mh, Mw)))
    NPD, NPH, NPW = (D + 1) // 2, (H + 1) // 2, (W + 1) // 2
    nstrips = (NPW + 63) // 64
    ngroups = -(-NPH // CY)
    nseg = -(-NPD // seg_out)
    flat = [b.reshape(Md, Mh * Mw) for b in bands]  # dense coefficient rows: the rows of a band's slice are one contiguous piece
    tlo = [(rec_lo[2 * j], rec_lo[2 * j + 1]) for j in range(HL)]
    thi = [(rec_hi[2 * j], rec_hi[2 * j + 1]) for j in range(HL)]
    y = np.full((D, H, W), np.nan)
    written = np.zeros((D, H, W), int)
    band_floats = 1 + (IY * Mw * 4 + 1023) // 1024 * 256  # the piece's 1-KiB requests (+ 1 float: an odd L/2 reads one pair too far)
    for s in range(nseg):
        PA = s * seg_out
        PB = min(NPD, PA + seg_out)
        nsl = PB - PA + HL - 1
        for g in range(ngroups):
            py0 = g * CY
            nrows = min(IY, Mh - py0)
            piece = nrows * Mw
            lds = np.full((nslots, 8, band_floats), np.nan)
            lds_id = np.full(nslots, -10**9)
            ib = 0

            def issue(t):
                nonlocal ib
                zc = PA + t
                assert zc < Md, "a segment never asks for a slice behind the bands"
                for b in range(8):
                    lds[ib, b, :] = np.nan
                    # lanes inside the piece move 16 bytes each; what they drag in behind the piece is whatever follows it in memory
                    n16 = -(-piece // 4) * 4
                    src = flat[b][zc, py0 * Mw: py0 * Mw + n16]
                    lds[ib, b, :len(src)] = src
                    if len(src) > piece:
                        lds[ib, b, piece:len(src)] = np.nan  # (poisoned here: no valid output may depend on it)
                lds_id[ib] = t
                ib = (ib + 1) % nslots

            ahead = nslots - 1
            for t in range(min(ahead, nsl)):
                issue(t)
            acc = {}
            slot = 0
            for t in range(nsl):
                assert lds_id[slot] == t
                if t + ahead < nsl:  # (requested while slice t is being read: a different slot)
                    assert ib != slot
                    issue(t + ahead)
                sl = lds[slot]
                R = t % HL
                for wave in range(nstrips):
                    lanes = np.arange(64)
                    pl = 2 * (lanes & 31) + (lanes >> 5) if st16 else lanes  # 16-byte stores: lanes l / l + 32 own neighbouring pairs
                    p = 64 * wave + pl
                    pc = np.minimum(p, Mw - HL)
                    # W pass first: (column 2p, 2p + 1) of the four (D, H) images, every staged row
                    wimg = np.zeros((2, 2, IY, 2, 64))
                    for dh in range(4):
                        lo_b, hi_b = sl[2 * dh], sl[2 * dh + 1]
                        for yy in range(IY):
                            for i in range(HL):
                                a_ = lo_b[yy * Mw + pc + i]
                                d_ = hi_b[yy * Mw + pc + i]
                                for r in range(2):
                                    wimg[dh >> 1, dh & 1, yy, r] += tlo[HL - 1 - i][r] * a_ + thi[HL - 1 - i][r] * d_
                    # H pass: rows (2 q, 2 q + 1) of the group
                    himg = np.zeros((2, CY, 2, 2, 64))  # [d][q][column c][row rr]
                    for d in range(2):
                        for q in range(CY):
                            for i in range(HL):
                                for rr in range(2):
                                    himg[d, q, :, rr] += tlo[HL - 1 - i][rr] * wimg[d, 0, q + i] + thi[HL - 1 - i][rr] * wimg[d, 1, q + i]
                    # D pass, rolling: slice t feeds the output slice pairs t - i (slot (R - i) mod HL), i = 0 starts a pair
                    A = acc.setdefault(wave, np.zeros((HL, CY, 2, 2, 2, 64)))  # [slot][q][c][rr][slice r of the pair]
                    for i in range(HL):
                        sl_i = (R - i + HL) % HL
                        if i == 0:
                            A[sl_i] = 0.0
                        for r in range(2):
                            A[sl_i, :, :, :, r] += tlo[HL - 1 - i][r] * himg[0] + thi[HL - 1 - i][r] * himg[1]
                    if t >= HL - 1:
                        P = PA + t - (HL - 1)
                        S = (R + 1) % HL
                        for r in range(2):
                            z = 2 * P + r
                            if z >= D:
                                continue
                            for q in range(CY):
                                if st16:
                                    # lanes l < 32 / l + 32 exchange: afterwards lane l holds columns 4 l' .. 4 l' + 3 of row 2 q, lane l + 32 of row 2 q + 1
                                    for half in range(2):
                                        n = 2 * (py0 + q) + half
                                        for l in range(32):
                                            c0 = 128 * wave + 4 * l
                                            if n >= H or c0 >= W:
                                                continue
                                            assert c0 + 3 < W, "the 16-byte path is taken for widths that are multiples of four only"
                                            lo_l, up_l = l, l + 32  # own the pairs 2 l, 2 l + 1 of the wave
                                            vals = [A[S, q, 0, half, r, lo_l], A[S, q, 1, half, r, lo_l],
                                                    A[S, q, 0, half, r, up_l], A[S, q, 1, half, r, up_l]]
                                            y[z, n, c0:c0 + 4] = vals
                                            written[z, n, c0:c0 + 4] += 1
                                else:
                                    for rr in range(2):
                                        n = 2 * (py0 + q) + rr
                                        if n >= H:
                                            continue
                                        for ln in range(64):
                                            xo = 2 * p[ln]
                                            if xo >= W:
                                                continue
                                            y[z, n, xo] = A[S, q, 0, rr, r, ln]
                                            written[z, n, xo] += 1
                                            if xo + 1 < W:
                                                y[z, n, xo + 1] = A[S, q, 1, rr, r, ln]
                                                written[z, n, xo + 1] += 1
                slot = (slot + 1) % nslots
    assert (written == 1).all(), "every output sample is stored exactly once"
    assert not np.isnan(y).any(), "a stored output depended on a coefficient behind a band's piece"
    return y


INV_CASES = [
    # (shape, wavelet, CY, seg_out, nslots)
    ((21, 19, 37), "db2", 4, 3, 3),
    ((12, 13, 140), "db2", 4, 100, 2),   # W % 4 == 0: 16-byte store path; two column strips
    ((10, 9, 260), "haar", 4, 2, 3),     # three strips
    ((14, 22, 40), "db3", 4, 4, 2),      # odd L/2: the W pass reads one coefficient pair too far, and never uses it
    ((17, 11, 36), "db4", 2, 5, 3),
    ((18, 17, 70), "db2", 2, 4, 3),      # the f64 instances' shape: two row pairs per workgroup for every filter length
    ((16, 15, 44), "db3", 2, 3, 2),
]


@pytest.mark.parametrize("case", INV_CASES)
def test_walk3_inv_model_vs_oracle(case):
    shape, wavelet, CY, seg_out, nslots = case
    rng = np.random.default_rng(len(wavelet) + shape[2])
    bank = O.filter_bank(wavelet)
    rlo, rhi = [float(v) for v in bank[2]], [float(v) for v in bank[3]]
    x = rng.standard_normal(shape)
    for mode in MODES:
        try:
            coeffs = O.wavedec3(x, wavelet, mode=mode, level=1)
        except RuntimeError:
            continue
        want = O.waverec3(coeffs, wavelet)
        bands = [coeffs[0]] + [coeffs[1][k] for k in KEYS[1:]]
        for st16 in ((False, True) if want.shape[2] % 4 == 0 else (False,)):
            got = walk3_inv_model(bands, rlo, rhi, want.shape, CY, seg_out, nslots, st16)
            np.testing.assert_allclose(got, want, rtol=0, atol=1e-12, err_msg=f"{case} {mode} st16={st16}")
