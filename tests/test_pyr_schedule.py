"""Host logic of the streaming 2-D analysis kernel (id 16): how a launch cuts the batch's rows into one chunk per persistent
workgroup (csrc/mifwt_dwt2_fwd_pyr.hip: pyr_schedule, exported as mifwt_dwt2_fwd_pyramid_schedule).  No GPU needed: the schedule is
computed on the host (256 CUs assumed when no device answers)."""
import ctypes

import numpy as np
import pytest

from ptwt_amd import _engine


def _descs(batch, H, W, L, nlev, mode="reflect"):
    descs = []
    h, w = H, W
    for _ in range(nlev):
        d = _engine.LevelDesc()
        d.ndim, d.dtype, d.mode, d.filt_len, d.batch = 2, 0, _engine.MODE_IDS[mode], L, batch
        ho, wo = (h + L - 1) // 2, (w + L - 1) // 2
        d.sig_extent[0], d.sig_extent[1] = h, w
        d.sig_stride[0], d.sig_stride[1], d.sig_stride[2] = h * w, w, 1
        d.coef_extent[0], d.coef_extent[1] = ho, wo
        d.approx_stride[0], d.approx_stride[1], d.approx_stride[2] = 4 * ho * wo, wo, 1
        d.detail_stride[0], d.detail_stride[1], d.detail_stride[2] = 4 * ho * wo, wo, 1
        descs.append(d)
        h, w = ho, wo
    return descs, h


def _schedule(batch, H, W, L, nlev, wgs=0):
    lib = _engine.load_library()
    lib.mifwt_dwt2_fwd_pyramid_schedule.restype = ctypes.c_int
    descs, hn = _descs(batch, H, W, L, nlev)
    arr = (ctypes.POINTER(_engine.LevelDesc) * nlev)(*[ctypes.pointer(d) for d in descs])
    out = (ctypes.c_uint32 * 400)()
    _engine.set_option(_engine.OPT_PYR_WGS, wgs)
    _engine.set_option(_engine.OPT_PYRAMID_MODE, 1)  # (wherever the kernel can run, not only where it is the fastest route)
    try:
        n = lib.mifwt_dwt2_fwd_pyramid_schedule(nlev, arr, out, 400)
    finally:
        _engine.set_option(_engine.OPT_PYR_WGS, 0)
        _engine.set_option(_engine.OPT_PYRAMID_MODE, 0)
    assert n >= 1, n
    return np.array(out[: n + 1], dtype=np.int64), hn


def _lags(L):
    cdiv = lambda a, b: 0 if a <= 0 else (a + b - 1) // b  # noqa: E731
    l2i = (L // 2) // 4 + 1
    l2 = max(cdiv(L - 2 + L // 2 - 2, 4), l2i)
    l3i = l2i + 1 + cdiv(L // 2 - 1, 2)
    l3 = l2 + 1 + cdiv(max(L // 2 - 1, L - 2 + L // 2 - 2), 2)
    return l2i, l2, l3i, l3


def _unit_time(lo, hi, L, Hs, nlev):
    """The kernel's step counts for rows [lo, hi) of the last level of one image, weighted as the scheduler does."""
    HL, HP = L - 2, L // 2
    rA, rB = {nlev: lo}, {nlev: hi}
    for l in range(nlev - 1, 0, -1):
        rA[l] = max(0, 2 * rA[l + 1] - HL)
        rB[l] = min(Hs[l], 2 * rB[l + 1])
    l2i, l2, l3i, l3 = _lags(L)
    top = lo == 0
    n1 = (rB[1] - rA[1] + HP - 1 + 3) // 4
    n = n1
    if nlev >= 2:
        n = max(n, (l2 if top else l2i) + (rB[2] - rA[2] + HP - 1 + 1) // 2)
    if nlev >= 3:
        n = max(n, (l3 if top else l3i) + rB[3] - rA[3] + HP - 1)
    return n1 + 0.45 * (n - n1) + 1.0 - (1.0 if top else 0.0)


def _chunk_times(cuts, hn, L, Hs, nlev):
    times = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        t, g = 0.0, a
        while g < b:
            img, lo = divmod(g, hn)
            rows = min(hn - lo, b - g)
            t += _unit_time(lo, lo + rows, L, Hs, nlev)
            g += rows
        times.append(t)
    return np.array(times)


@pytest.mark.parametrize("batch", [1, 3, 16, 48, 64, 65, 72, 96, 100, 128, 200, 256, 1000])
def test_chunks_cover_the_rows_and_take_equal_modelled_time(batch):
    H = W = 1024
    L, nlev = 8, 3
    cuts, hn = _schedule(batch, H, W, L, nlev)
    Hs = [H]
    for _ in range(nlev):
        Hs.append((Hs[-1] + L - 1) // 2)
    assert hn == Hs[nlev] == 134
    assert cuts[0] == 0 and cuts[-1] == batch * hn and np.all(np.diff(cuts) > 0)
    assert len(cuts) - 1 <= 256
    # no unit shorter than 8 rows unless it is a whole image
    for a, b in zip(cuts[:-1], cuts[1:]):
        for edge in (a, b):
            r = edge % hn
            assert r == 0 or 8 <= r <= hn - 8, (batch, edge, r)
    t = _chunk_times(cuts, hn, L, Hs, nlev)
    if batch * hn >= 256 * 16:
        # (fewer chunks than CUs where one more cut per image would cost more — a prologue and a drain — than it spreads)
        assert 192 <= len(cuts) - 1 <= 256
        assert t.max() <= 1.06 * t.mean() + 1.0, (batch, t.max(), t.mean())
        # ... and never worse than a third over the ideal spread of the one-unit-per-image times over 256 workgroups
        ideal = batch * _unit_time(0, hn, L, Hs, nlev) / 256
        assert t.max() <= 1.34 * ideal + 8.0, (batch, t.max(), ideal)


def test_config2_is_four_units_per_image():
    """64 x 1024^2 db4 level 3 on 256 CUs: the cut the round-5 wall clocks asked for — first / inner / last segment of an image
    34 / 32..33 / 35..36 rows of the 134 (equal rows: 34 / 34 / 34 / 32, the inner segments ended 10 us after the last one)."""
    cuts, hn = _schedule(64, 1024, 1024, 8, 3)
    assert len(cuts) == 257
    per_image = cuts.reshape(-1)[:-1].reshape(64, 4) - (np.arange(64) * hn)[:, None]
    assert np.all(per_image[:, 0] == 0)
    rows = np.diff(np.concatenate([per_image, np.full((64, 1), hn)], axis=1), axis=1)
    assert np.all(rows == rows[0])
    assert 33 <= rows[0][0] <= 35 and 31 <= rows[0][1] <= 33 and 31 <= rows[0][2] <= 33 and 34 <= rows[0][3] <= 37, rows[0]


@pytest.mark.parametrize("wgs", [1, 2, 3, 7, 50])
def test_forced_chunk_counts(wgs):
    cuts, hn = _schedule(5, 520, 600, 8, 3, wgs=wgs)
    assert cuts[0] == 0 and cuts[-1] == 5 * hn and len(cuts) - 1 <= wgs
    if wgs <= 7:
        assert len(cuts) - 1 == wgs


@pytest.mark.parametrize("L,nlev", [(2, 3), (4, 2), (6, 3), (8, 1), (8, 2)])
def test_other_filters_and_level_counts(L, nlev):
    for batch in (7, 64, 90):
        cuts, hn = _schedule(batch, 640, 768, L, nlev)
        assert cuts[0] == 0 and cuts[-1] == batch * hn and np.all(np.diff(cuts) > 0)
