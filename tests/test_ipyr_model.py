"""CPU model of the streaming multi-level 2-D synthesis kernel (csrc/mifwt_dwt2_inv_pyr.hip, kernel id 22): the row ranges of a
row segment at every level, the sub-step schedule of the three wave roles (lags D2 / T1), the staging entries the loader waves fill
(which row of which level lands in which entry, how far ahead), the LDS rings between the levels, the rolling vertical pass and the
horizontal polyphase pass — with every LDS read checked for "written in an EARLIER sub-step and not overwritten since" (one
workgroup barrier per sub-step is the only synchronisation the kernel has).  The result is compared with the oracle's waverec2
(src/ptwt/conv_transform_2.py:222-249).  The constants computed here (``ipyr_schedule``) are the ones the kernel uses."""
import numpy as np
import pytest

from oracle import fwt_oracle as O

RING1, RING2 = 16, 8  # rows of the approximation rings of level 1 / level 2


def ipyr_schedule(L, nlev):
    """(D2, T1): level 2 runs D2 STEPS (two sub-steps each) behind level 3, level 1 runs T1 SUB-STEPS behind sub-step 0."""
    HL = L // 2
    D2 = HL if nlev >= 3 else 0
    T1 = 0 if nlev == 1 else 2 * D2 + HL + 1
    return D2, T1


def seg_ranges(L, nlev, y0, y1):
    """Coefficient rows [a_l, b_l] (inclusive) of level l = 1 .. nlev a segment of output rows [y0, y1) consumes (y0 a multiple of 8)."""
    HL = L // 2
    a, b = [0] * (nlev + 1), [0] * (nlev + 1)
    a[1], b[1] = y0 // 2, (y1 + 1) // 2 - 1 + HL - 1
    for l in range(2, nlev + 1):
        a[l], b[l] = a[l - 1] // 2, b[l - 1] // 2 + HL - 1
    return a, b


class Lds:
    """Rows in LDS with the sub-step they were written in; reads must see a row written strictly earlier."""

    def __init__(self):
        self.rows = {}

    def write(self, key, tag, data, t):
        self.rows[key] = (tag, np.array(data, dtype=np.float64), t)

    def read(self, key, tag, t):
        assert key in self.rows, f"sub-step {t}: slot {key} was never written (want {tag})"
        got_tag, data, tw = self.rows[key]
        assert got_tag == tag, f"sub-step {t}: slot {key} holds {got_tag}, want {tag}"
        assert tw < t, f"sub-step {t}: slot {key} ({tag}) was written in the same or a later sub-step ({tw})"
        return data


def hsyn(lo, hi, c_lo, c_hi, nout):
    """Horizontal synthesis of one coefficient row pair (lo branch, hi branch) -> nout samples: y[2q + r] = sum_t g[2 (HL-1-t) + r] c[q + t]."""
    HL = len(lo) // 2
    out = np.zeros(2 * ((nout + 1) // 2) + 2)
    for q in range((nout + 1) // 2):
        for t in range(HL):
            j = HL - 1 - t
            for r in range(2):
                out[2 * q + r] += lo[2 * j + r] * c_lo[q + t] + hi[2 * j + r] * c_hi[q + t]
    return out[:nout]


def simulate(coeffs, lo, hi, out_hw, seg_rows, nbuf=5, fast=True):
    """coeffs = [aa_N, (da_N, ad_N, dd_N), ..., (da_1, ad_1, dd_1)] — one image in the oracle's (pywt) order: H = 'da', V = 'ad', D = 'dd'."""
    L = len(lo)
    HL = L // 2
    nlev = len(coeffs) - 1
    # bands[l][b]: level l = 1 (finest) .. nlev; b = 0 only for the coarsest
    bands = [None] * (nlev + 1)
    for l in range(1, nlev + 1):
        det = coeffs[nlev + 1 - l]
        bands[l] = [coeffs[0] if l == nlev else None, det[1], det[0], det[2]]  # engine order: aa, ad, da, dd
    Mh = [0] + [bands[l][1].shape[0] for l in range(1, nlev + 1)]
    Mw = [0] + [bands[l][1].shape[1] for l in range(1, nlev + 1)]
    H, W = out_hw
    Nh = [0, H] + [Mh[l - 1] for l in range(2, nlev + 1)]  # output extents of level l
    Nw = [0, W] + [Mw[l - 1] for l in range(2, nlev + 1)]
    for l in range(1, nlev + 1):
        assert Nh[l] <= 2 * Mh[l] - L + 2 and Nw[l] <= 2 * Mw[l] - L + 2
    D2, T1 = ipyr_schedule(L, nlev)
    ahead = nbuf - 2
    # fast warm-up: the entries of the sub-steps before level 1 starts (their level-2 / level-3 rows only) have slots of their own,
    # all requested when the workgroup starts; the ring of nbuf entries begins with entry TW
    TW = T1 if fast else 0
    y = np.full((H, W), np.nan)
    assert seg_rows % 8 == 0
    for y0 in range(0, H, seg_rows):
        y1 = min(H, y0 + seg_rows)
        a, b = seg_ranges(L, nlev, y0, y1)
        for l in range(1, nlev + 1):
            assert 0 <= a[l] and b[l] <= Mh[l] - 1, (l, a, b, Mh)
        nsub1 = (b[1] - a[1] + 2) // 2
        nsub = T1 + nsub1
        lds = Lds()
        issued = set()

        def slot(t):
            return ("wu", t) if t < TW else ("st", (t - TW) % nbuf)

        def entry_rows(t):
            """What the loader puts into staging entry t: list of (slot key, tag, data)."""
            out = []
            if t >= T1:
                for j in range(2):
                    r = a[1] + 2 * (t - T1) + j
                    for bnd in range(0 if nlev == 1 else 1, 4):
                        alive = a[1] <= r <= b[1]
                        out.append(((*slot(t), 1, bnd, j), ("L1", bnd, r) if alive else ("dead",), bands[1][bnd][r] if alive else np.zeros(Mw[1])))
            if nlev >= 2:
                r = a[2] + t - 2 * D2
                for bnd in range(0 if nlev == 2 else 1, 4):
                    alive = a[2] <= r <= b[2]
                    if t < TW and t < 2 * D2:
                        continue  # (no slot: level 2 has not started)
                    out.append(((*slot(t), 2, bnd, 0), ("L2", bnd, r) if alive else ("dead",), bands[2][bnd][r] if alive else np.zeros(Mw[2])))
            if nlev >= 3:
                r = a[3] + t // 2
                for i in range(2):
                    bnd = 2 * (t & 1) + i
                    alive = a[3] <= r <= b[3]
                    out.append(((*slot(t), 3, i, 0), ("L3", bnd, r) if alive else ("dead",), bands[3][bnd][r] if alive else np.zeros(Mw[3])))
            return out

        def issue(t, now):
            assert t not in issued
            issued.add(t)
            for key, tag, data in entry_rows(t):
                lds.write(key, tag, data, now)

        for t in range(min(TW + ahead, nsub)):
            issue(t, -1)
        acc = {l: {} for l in range(1, nlev + 1)}  # acc[l][p] = [rows 2p and 2p+1 of the level's output, partial sums]
        nfed = {l: 0 for l in range(1, nlev + 1)}

        def feed(l, r, t, src):
            """Level l consumes its coefficient row r (bands via src(bnd))."""
            c = [src(bnd) for bnd in range(4)]
            nout = Nw[l]
            v_lo = hsyn(lo, hi, c[0], c[1], nout)  # vertical-lo image row (bands aa, ad)
            v_hi = hsyn(lo, hi, c[2], c[3], nout)
            for i in range(HL):
                p = r - i
                j = HL - 1 - i
                rows = acc[l].setdefault(p, np.zeros((2, nout)))
                for rr in range(2):
                    rows[rr] += lo[2 * j + rr] * v_lo + hi[2 * j + rr] * v_hi
            nfed[l] += 1
            p = r - (HL - 1)
            done = acc[l].pop(p)
            for q in [k for k in acc[l] if k < p]:
                del acc[l][q]
            return p, done

        for t in range(nsub):
            # (barrier t)  the loader refills the buffer nobody reads any more
            # level 3 / level 2 at odd sub-steps
            if t & 1:
                s = t // 2
                if nlev >= 3:
                    r3 = a[3] + s
                    if r3 <= b[3]:
                        def src3(bnd, r3=r3, t=t):
                            return lds.read((*slot(t - 1 + (bnd >> 1)), 3, bnd & 1, 0), ("L3", bnd, r3), t)
                        p, done = feed(3, r3, t, src3)
                        if p >= a[3]:
                            for rr in range(2):
                                lds.write(("ring2", (2 * p + rr) % RING2), ("aa2", 2 * p + rr), done[rr], t)
                if nlev >= 2 and s >= D2:
                    for j in range(2):
                        r2 = a[2] + 2 * (s - D2) + j
                        if r2 > b[2]:
                            continue
                        def src2(bnd, r2=r2, t=t, j=j):
                            if bnd == 0 and nlev >= 3:
                                return lds.read(("ring2", r2 % RING2), ("aa2", r2), t)
                            return lds.read((*slot(t - 1 + j), 2, bnd, 0), ("L2", bnd, r2), t)
                        p, done = feed(2, r2, t, src2)
                        if p >= a[2]:
                            for rr in range(2):
                                lds.write(("ring1", (2 * p + rr) % RING1), ("aa1", 2 * p + rr), done[rr], t)
            if t >= T1:
                for j in range(2):
                    r1 = a[1] + 2 * (t - T1) + j
                    if r1 > b[1]:
                        continue
                    def src1(bnd, r1=r1, t=t, j=j):
                        if bnd == 0 and nlev >= 2:
                            return lds.read(("ring1", r1 % RING1), ("aa1", r1), t)
                        return lds.read((*slot(t), 1, bnd, j), ("L1", bnd, r1), t)
                    p, done = feed(1, r1, t, src1)
                    if p >= a[1]:
                        for rr in range(2):
                            if y0 <= 2 * p + rr < y1:
                                y[2 * p + rr] = done[rr]
            if t >= TW and t + ahead < nsub:
                issue(t + ahead, t)  # lands before barrier t + ahead; overwrites entry t + ahead - nbuf = t - 2
    assert not np.isnan(y).any()
    return y


CASES = [("db4", 3, (200, 190), 64), ("db4", 3, (256, 260), 64), ("db2", 3, (130, 150), 32), ("haar", 3, (128, 96), 32), ("db3", 3, (150, 131), 48),
         ("db5", 3, (260, 210), 64), ("db5", 2, (131, 150), 48), ("sym5", 1, (80, 70), 40), ("db4", 2, (120, 100), 40), ("db3", 2, (97, 110), 32), ("db2", 1, (70, 66), 24), ("db4", 1, (64, 64), 64), ("haar", 2, (64, 80), 16)]


@pytest.mark.parametrize("wavelet,level,shape,seg_rows", CASES)
def test_schedule_model_matches_oracle(wavelet, level, shape, seg_rows):
    rng = np.random.default_rng(7)
    x = rng.standard_normal(shape)
    for mode in ("reflect", "zero"):
        coeffs = O.wavedec2(x, wavelet, mode=mode, level=level)
        bank = O.filter_bank(wavelet)
        for fast in (True, False):
            got = simulate(coeffs, bank[2], bank[3], shape, seg_rows, fast=fast)
            want = O.waverec2(coeffs, wavelet)[: shape[0], : shape[1]]
            assert np.abs(got - want).max() < 1e-11 * max(1.0, np.abs(want).max())
        # random coefficients (not the image of an analysis): synthesis alone
        rc = [rng.standard_normal(coeffs[0].shape)] + [tuple(rng.standard_normal(b.shape) for b in lv) for lv in coeffs[1:]]
        got = simulate(rc, bank[2], bank[3], shape, seg_rows)
        want = O.waverec2(rc, wavelet)[: shape[0], : shape[1]]
        assert np.abs(got - want).max() < 1e-11 * max(1.0, np.abs(want).max())
