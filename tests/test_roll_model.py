"""CPU model of the rolling two-level analysis kernel's bookkeeping (csrc/mifwt_dwt2_fwd_roll.hip): column strips with
the shifted level-1 window, row segments (the first top-aligned, the others bottom-aligned), the prologue, the two
fixed-slot LDS windows with copy-down of the last L-2 rows, and level 2's boundary extension read from ACTUAL level-1
rows through the index map.  The model tracks which row every window slot holds and asserts each read hits the row the
maths needs; its output is compared with the oracle's two-level wavedec2.  (The GPU tests compare the kernel itself
bit-for-bit with the per-level kernels; this test pins the index logic where no GPU is available.)"""
import numpy as np
import pytest

from oracle import fwt_oracle as O


def _ext(i, n, mode):
    return int(O.ext_index(np.asarray([i]), n, mode)[0])


def _model(x, lo, hi, mode, seg):
    L = len(lo)
    HL, C1 = L - 2, 64
    T2C = (C1 - HL) // 2
    OC1, C0 = 2 * T2C, 2 * C1 + HL
    S2, S1, S0 = 8, 16, 32
    RH, RL = S0 + HL, S1 + HL
    H0, W0 = x.shape
    H1, W1 = (H0 + L - 1) // 2, (W0 + L - 1) // 2
    H2, W2 = (H1 + L - 1) // 2, (W1 + L - 1) // 2
    assert H1 >= 32 and W1 >= 64 and H2 >= 9
    seg = min((seg + 7) // 8 * 8, (H2 - 1) // 8 * 8)  # roll_segment()
    nseg = -(-H2 // seg)
    assert nseg >= 2
    d1 = np.full((3, H1, W1), np.nan)
    o2 = np.full((4, H2, W2), np.nan)
    wr1 = np.zeros((H1, W1), int)
    wr2 = np.zeros((H2, W2), int)
    kcols = np.arange(C1)
    for sg in range(nseg):
        jb = H2 - (nseg - 1 - sg) * seg
        ja = jb - seg if sg > 0 else 0
        nsteps = (jb - ja + S2 - 1) // S2
        own_lo, own_hi = 2 * ja, min(2 * jb, H1)
        for tc in range(-(-W2 // T2C)):
            k2_0 = tc * T2C
            s1c = min(max(2 * k2_0 - HL, 0), W1 - C1)
            c_first = 2 * s1c - HL
            cols_in2 = 2 * k2_0 - HL >= 0 and 2 * k2_0 + OC1 <= W1
            cmap = np.array([_ext(c_first + c, W0, mode) for c in range(C0)])
            hwin = np.zeros((RH, C1, 2))
            hid = np.full(RH, -10**9)
            awin = np.zeros((RL, C1))      # level-1 approximation rows
            hlwin = np.zeros((RL, 32, 2))  # ... and their level-2 horizontal (lo, hi) image
            aid = np.full(RL, -10**9)
            kk = np.arange(T2C)
            live_c = k2_0 + kk < W2
            cidx = np.full((L, T2C), -1)
            for p in range(L):
                for q in range(T2C):
                    if live_c[q]:
                        m = _ext(2 * (k2_0 + q) - HL + p, W1, mode)
                        cidx[p, q] = -1 if m < 0 else m - s1c
                        if cols_in2:
                            assert cidx[p, q] == 2 * q + p
                        assert cidx[p, q] < 64

            def h1(r, slot):
                dead = r < -HL or r >= 2 * H1
                m = -1 if dead else _ext(r, H0, mode)
                raw = np.zeros(C0)
                if m >= 0:
                    raw = np.where(cmap >= 0, x[m, np.maximum(cmap, 0)], 0.0)
                a = sum(lo[t] * raw[(L - 1 - t) + 2 * kcols] for t in range(L))
                b = sum(hi[t] * raw[(L - 1 - t) + 2 * kcols] for t in range(L))
                hwin[slot, :, 0], hwin[slot, :, 1] = a, b
                hid[slot] = r if not dead else -10**9 + 1

            def v1_h2(m1r, hslots, aslot, store):
                aa = da = ad = dd = 0
                for t in range(L):
                    s = hslots[t]
                    if 0 <= m1r < H1:
                        assert hid[s] == 2 * m1r + 1 - t, (hid[s], m1r, t)
                    hv = hwin[s]
                    aa = aa + lo[t] * hv[:, 0]
                    da = da + hi[t] * hv[:, 0]
                    ad = ad + lo[t] * hv[:, 1]
                    dd = dd + hi[t] * hv[:, 1]
                awin[aslot] = aa
                aid[aslot] = m1r
                if store and own_lo <= m1r < own_hi:
                    m1c = s1c + kcols
                    own = (m1c >= 2 * k2_0) & (m1c < 2 * k2_0 + OC1)
                    d1[0, m1r, m1c[own]], d1[1, m1r, m1c[own]], d1[2, m1r, m1c[own]] = ad[own], da[own], dd[own]
                    wr1[m1r, m1c[own]] += 1
                acc = np.zeros((T2C, 2))
                for p in range(L):
                    val = np.where(cidx[p] >= 0, awin[aslot, np.maximum(cidx[p], 0)], 0.0)
                    acc[:, 0] += lo[L - 1 - p] * val
                    acc[:, 1] += hi[L - 1 - p] * val
                hlwin[aslot, :T2C] = acc

            # prologue: level-0 rows [4 ja - 3 HL, 4 ja) -> slots (last HL -> [0, HL), first 2 HL -> [HL, 3 HL))
            for q in range(3 * HL):
                h1(4 * ja - 3 * HL + q, q + HL if q < 2 * HL else q - 2 * HL)
            for il in range(HL):
                hs = [(lambda q: q + HL if q < 2 * HL else q - 2 * HL)(2 * il + (L - 1) - t) for t in range(L)]
                v1_h2(2 * ja - HL + il, hs, il, False)
            for st in range(nsteps):
                j = ja + S2 * st
                for i in range(S0):
                    h1(4 * j + i, HL + i)
                if st > 0:  # copy-down of the approximation window
                    for t in range(HL):
                        awin[t], hlwin[t], aid[t] = awin[S1 + t], hlwin[S1 + t], aid[S1 + t]
                for i in range(S1):
                    m1r = 2 * j + i
                    v1_h2(m1r, [2 * i + (L - 1) - t for t in range(L)], HL + i, True)
                for t in range(HL):  # copy-down of the h-window
                    hwin[t], hid[t] = hwin[S0 + t].copy(), hid[S0 + t]
                for j2 in range(j, min(j + S2, jb)):
                    acc = np.zeros((4, T2C))
                    for t in range(L):
                        e = _ext(2 * j2 + 1 - t, H1, mode)
                        if e < 0:
                            continue
                        sl = e - (2 * j - HL)
                        assert 0 <= sl < RL and aid[sl] == e, (sl, aid[sl] if 0 <= sl < RL else None, e, j2, j)
                        hv = hlwin[sl, :T2C]
                        acc += np.stack([lo[t] * hv[:, 0], lo[t] * hv[:, 1], hi[t] * hv[:, 0], hi[t] * hv[:, 1]])
                    cols = k2_0 + kk[live_c]
                    o2[:, j2, cols] = acc[:, live_c]
                    wr2[j2, cols] += 1
    assert (wr1 == 1).all() and (wr2 == 1).all()
    return d1, o2


@pytest.mark.parametrize("wavelet,shape,seg", [("db4", (150, 140), 8), ("db4", (151, 139), 24), ("db2", (130, 171), 32),
                                               ("haar", (128, 130), 8), ("db3", (167, 255), 16), ("db4", (262, 129), 24)])
@pytest.mark.parametrize("mode", ["reflect", "zero", "constant", "symmetric"])
def test_rolling_pair_kernel_index_logic(wavelet, shape, seg, mode):
    rng = np.random.default_rng(0)
    fb = O.filter_bank(wavelet)
    x = rng.standard_normal(shape)
    d1, o2 = _model(x, np.asarray(fb[0]), np.asarray(fb[1]), mode, seg)
    want = O.wavedec2(x[None], wavelet, mode=mode, level=2)
    # want = (cA2, (H, V, D)_2, (H, V, D)_1) with H = da, V = ad, D = dd; d1 = (ad, da, dd), o2 = (aa, ad, da, dd)
    pairs = [(o2[0], want[0][0]), (o2[2], want[1][0][0]), (o2[1], want[1][1][0]), (o2[3], want[1][2][0]),
             (d1[1], want[2][0][0]), (d1[0], want[2][1][0]), (d1[2], want[2][2][0])]
    for got, ref in pairs:
        assert np.abs(got - ref).max() < 1e-12
