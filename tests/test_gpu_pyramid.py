"""GPU parity tests (``-m gpu``) of the multi-level launch ``mifwt_dwt2_fwd_pyramid`` (kernel id 16: up to three 2-D analysis
levels per launch, solo column strips + loader wave, mifwt_dwt2_fwd_pyr.hip) against the fp64 numpy oracle.

Tolerance: fp32 <= 1e-6 norm-wise per sub-band vs the fp64 oracle.  Every case asserts that the pyramid kernel is
the one that ran (``_engine.level_events``), so a silent per-level fallback cannot pass.
"""
import numpy as np
import pytest
import torch

import ptwt_amd
from oracle import fwt_oracle as O
from ptwt_amd import _engine
from tests import _golden as G

pytestmark = pytest.mark.gpu

TOL32 = 1e-6
MODES = ["reflect", "zero", "constant", "symmetric"]


def dev():
    return torch.device("cuda:0")


@pytest.fixture(autouse=True, params=[16, 1], ids=["rows16", "dense"])
def _pyramid_wherever_it_can_run(request):
    """Auto mode keeps the kernel to planes of 448 .. ~2560 columns (where it is the fastest route); the parity cases here are
    mostly smaller or wider, so they switch to "wherever the kernel can run" (MIFWT_OPT_PYRAMID_MODE 1).  Every case runs twice: with
    the streaming kernel's planes on 16-byte aligned rows (16-byte stores after a lane-pair exchange) and with dense planes (the
    default: 8-byte stores wherever a width is not a multiple of four)."""
    old = _engine.PYRAMID_ROW_ALIGN
    _engine.PYRAMID_ROW_ALIGN = request.param
    _engine.set_option(_engine.OPT_PYRAMID_MODE, 1)
    yield
    _engine.PYRAMID_ROW_ALIGN = old
    _engine.set_option(_engine.OPT_PYRAMID_MODE, 0)


def run_traced(fn):
    """Run ``fn`` with per-level events on; returns (result, [kernel ids of the launches])."""
    _engine.level_events = []
    try:
        out = fn()
        torch.cuda.synchronize()
        kids = [e[1] for e in _engine.level_events]
    finally:
        _engine.level_events = None
    return out, kids


def check(x, wavelet, mode, level, want_kids=None, seg_rows=0):
    if seg_rows:
        _engine.set_option(_engine.OPT_PAIR_ROWS, seg_rows)
    try:
        got, kids = run_traced(lambda: ptwt_amd.wavedec2(x.to(dev()), wavelet, mode=mode, level=level))
    finally:
        if seg_rows:
            _engine.set_option(_engine.OPT_PAIR_ROWS, 0)
    want = O.wavedec2(x.numpy().astype(np.float64), wavelet, mode=mode, level=level)
    if want_kids is not None:
        assert kids == want_kids, f"launches {kids}, expected {want_kids}"
    gf, wf = G.flatten_coeffs(got), G.flatten_coeffs(want)
    assert [n for n, _ in gf] == [n for n, _ in wf]
    worst = 0.0
    for (n, a), (_, b) in zip(gf, wf):
        assert tuple(a.shape) == tuple(np.shape(b)), n
        err = G.relerr(a.cpu().numpy(), b)
        worst = max(worst, err)
        assert err < TOL32, f"{wavelet} {mode} L{level} {tuple(x.shape)} {n}: rel err {err:.3e}"
    return worst


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("wavelet", ["haar", "db2", "db3", "db4"])
def test_pyramid_three_levels_vs_oracle(wavelet, mode):
    g = torch.Generator().manual_seed(11)
    x = torch.randn(3, 200, 328, generator=g, dtype=torch.float32)  # several strips, one workgroup per image
    check(x, wavelet, mode, 3, want_kids=[_engine.KID_PYRAMID])


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("level", [1, 2])
def test_pyramid_one_and_two_levels(level, mode):
    g = torch.Generator().manual_seed(12)
    x = torch.randn(2, 131, 260, generator=g, dtype=torch.float32)  # odd height: the extra pad row of the reference
    check(x, "db4", mode, level, want_kids=[_engine.KID_PYRAMID])


@pytest.mark.parametrize("wavelet", ["haar", "db2", "db3", "db4"])
def test_pyramid_row_segments(wavelet):
    """Several row segments per image (prologues, top-aligned mirror rows, last segment takes the rest)."""
    g = torch.Generator().manual_seed(13)
    x = torch.randn(2, 300, 264, generator=g, dtype=torch.float32)
    for mode in ("reflect", "symmetric", "zero"):
        check(x, wavelet, mode, 3, want_kids=[_engine.KID_PYRAMID], seg_rows=8)
        check(x, wavelet, mode, 2, want_kids=[_engine.KID_PYRAMID], seg_rows=16)


def test_pyramid_store_paths_ragged_widths():
    """The streaming kernel's stores against the oracle on level-1 / level-2 widths of every residue modulo four (a ragged last lane
    stores a single column), with short row chunks (odd ownership boundaries); `mifwt_launch_count` pins that the kernel ran."""
    g = torch.Generator().manual_seed(21)
    for width in (512, 514, 516, 518, 1024, 1030):
        x = torch.randn(2, 140, width, generator=g, dtype=torch.float32)
        for wavelet, level in (("haar", 3), ("db2", 2), ("db3", 3), ("db4", 3), ("db4", 1)):
            n8 = _engine.launch_count(_engine.VARIANT_FWD_PYR_ST8)
            check(x, wavelet, "symmetric", level, want_kids=[_engine.KID_PYRAMID])
            check(x, wavelet, "zero", level, want_kids=[_engine.KID_PYRAMID], seg_rows=8)
            assert _engine.launch_count(_engine.VARIANT_FWD_PYR_ST8) - n8 == 2


def test_pyramid_many_strips_two_groups():
    """A plane wide enough for more strips than one workgroup has compute waves."""
    g = torch.Generator().manual_seed(14)
    x = torch.randn(1, 96, 2304, generator=g, dtype=torch.float32)
    for mode in ("reflect", "constant"):
        check(x, "db4", mode, 3, want_kids=[_engine.KID_PYRAMID])
        check(x, "db2", mode, 3, want_kids=[_engine.KID_PYRAMID])


def test_pyramid_then_more_levels():
    """Five levels: three in the pyramid launch, the rest by the per-level / pair kernels."""
    g = torch.Generator().manual_seed(15)
    x = torch.randn(2, 512, 512, generator=g, dtype=torch.float32)
    _, kids = run_traced(lambda: ptwt_amd.wavedec2(x.to(dev()), "db4", level=5))
    assert kids[0] == _engine.KID_PYRAMID and len(kids) >= 2
    check(x, "db4", "reflect", 5)


def test_pyramid_small_and_ragged_planes():
    g = torch.Generator().manual_seed(16)
    for shape in [(2, 64, 64), (1, 97, 132), (3, 70, 528), (1, 1000, 36), (2, 129, 1028)]:
        x = torch.randn(*shape, generator=g, dtype=torch.float32)
        for wavelet in ("haar", "db4"):
            for level in (1, 2, 3):
                check(x, wavelet, "reflect", level)
                check(x, wavelet, "symmetric", level)


def test_pyramid_config2_full_size_all_64_images_vs_device_f64_and_16_images_vs_numpy_oracle():
    """BASELINE config 2 (64 x 1024^2 db4 level 3): every image of the batch against an on-device fp64 run of the per-level tile
    kernels (already pinned against the oracle and the goldens), 4 images against the numpy oracle itself."""
    g = torch.Generator().manual_seed(17)
    x = torch.randn(64, 1024, 1024, generator=g, dtype=torch.float32)
    xd = x.to(dev())
    got, kids = run_traced(lambda: ptwt_amd.wavedec2(xd, "db4", level=3))
    assert kids == [_engine.KID_PYRAMID]
    ref = ptwt_amd.wavedec2(xd.double(), "db4", level=3)
    for (n, a), (_, b) in zip(G.flatten_coeffs(got), G.flatten_coeffs(ref)):
        num = (a.double() - b).flatten(1).norm(dim=1)
        den = b.flatten(1).norm(dim=1)
        assert float((num / den).max()) < TOL32, n  # per image
    # sixteen images against the numpy oracle itself (one vectorised oracle call per group of four)
    for i0 in (0, 20, 40, 60):
        want = O.wavedec2(x[i0 : i0 + 4].numpy().astype(np.float64), "db4", level=3)
        for (n, a), (_, b) in zip(G.flatten_coeffs(got), G.flatten_coeffs(want)):
            for k in range(4):
                assert G.relerr(a[i0 + k].cpu().numpy(), b[k]) < TOL32, (i0 + k, n)


def test_pyramid_batch_beyond_two_gib():
    """600 x 1024^2 fp32: 2.5 GB of input, 2.5 GB of coefficients — image byte offsets beyond 2^31 through the streaming kernel (64-bit
    image bases, 32-bit offsets inside an image).  The oracle on the first, one in the middle and the last image; every image of the
    big batch bit-identical to the same image transformed in a 4-image batch (same kernel, small offsets)."""
    g = torch.Generator(device=dev()).manual_seed(19)
    xd = torch.empty(600, 1024, 1024, device=dev())
    for i in range(0, 600, 50):
        xd[i : i + 50] = torch.randn(50, 1024, 1024, device=dev(), generator=g)
    got, kids = run_traced(lambda: ptwt_amd.wavedec2(xd, "db4", level=3))
    assert kids == [_engine.KID_PYRAMID]
    for i in (0, 311, 599):
        want = O.wavedec2(xd[i : i + 1].cpu().numpy().astype(np.float64), "db4", level=3)
        for (n, a), (_, b) in zip(G.flatten_coeffs(got), G.flatten_coeffs(want)):
            assert G.relerr(a[i : i + 1].cpu().numpy(), b) < TOL32, (i, n)
    for i0 in range(0, 600, 40):
        small = ptwt_amd.wavedec2(xd[i0 : i0 + 4], "db4", level=3)
        for (n, a), (_, b) in zip(G.flatten_coeffs(got), G.flatten_coeffs(small)):
            assert torch.equal(a[i0 : i0 + 4], b), (i0, n)
    rec, kids = run_traced(lambda: ptwt_amd.waverec2(got, "db4"))  # the streaming synthesis launch over the same offsets
    assert kids == [_engine.KID_INV_PYRAMID]
    for i0 in range(0, 600, 100):
        assert float((rec[i0 : i0 + 100] - xd[i0 : i0 + 100]).abs().max()) < 2e-5


def test_pyramid_linearity_and_roundtrip_full_size():
    g = torch.Generator().manual_seed(18)
    x = torch.randn(8, 1024, 1024, generator=g, dtype=torch.float32).to(dev())
    y = torch.randn(8, 1024, 1024, generator=g, dtype=torch.float32).to(dev())
    cx = ptwt_amd.wavedec2(x, "db4", level=3)
    cy = ptwt_amd.wavedec2(y, "db4", level=3)
    cz = ptwt_amd.wavedec2(2.0 * x - 0.5 * y, "db4", level=3)
    for (n, a), (_, b), (_, c) in zip(G.flatten_coeffs(cx), G.flatten_coeffs(cy), G.flatten_coeffs(cz)):
        assert float(((2.0 * a - 0.5 * b) - c).norm() / c.norm()) < 2e-6, n
    rec = ptwt_amd.waverec2(cx, "db4")
    assert float((rec - x).abs().max()) < 1e-5


def test_pyramid_declines_what_it_cannot_serve():
    lib = _engine.load_library()
    taps = ptwt_amd._fwt.host_taps("db4")[:2]
    # auto mode: only where it is the fastest route — planes of 448 .. ~2560 columns (one or two column groups)
    _engine.set_option(_engine.OPT_PYRAMID_MODE, 0)
    for shape, served in (((4, 1024, 1024), True), ((4, 600, 512), True), ((4, 256, 256), False), ((2, 512, 2048), True), ((2, 512, 4096), False)):
        got = _engine.ENGINE.analysis_pyramid(torch.randn(*shape, device=dev()), *taps, _engine.MODE_IDS["reflect"], 3)
        assert (got is not None) == served, shape
    _engine.set_option(_engine.OPT_PYRAMID_MODE, 1)
    x = torch.randn(2, 640, 640, device=dev())  # (a plane too big for the small-plane kernel, which does serve periodic)
    # (round 5: the periodic extension is served one level per launch — the rings of a second level would need the plane's far side)
    got = _engine.ENGINE.analysis_pyramid(x, *ptwt_amd._fwt.host_taps("db4")[:2], _engine.MODE_IDS["periodic"], 3)
    assert got is not None and len(got) == 1
    got = _engine.ENGINE.analysis_pyramid(x, *ptwt_amd._fwt.host_taps("db6")[:2], _engine.MODE_IDS["reflect"], 3)
    assert got is None  # twelve taps
    x = torch.randn(2, 128, 128, device=dev())
    assert _engine.ENGINE.analysis_pyramid(x.double(), *ptwt_amd._fwt.host_taps("db4")[:2], _engine.MODE_IDS["reflect"], 3) is None
    # results of unsupported geometries still come from the other kernels
    got = ptwt_amd.wavedec2(torch.randn(2, 640, 640, device=dev()), "db4", level=2, mode="periodic")
    assert got[0].shape[-1] == 165
    del lib


@pytest.mark.parametrize("mode", MODES)
def test_pyramid_rows_of_any_length_and_alignment(mode):
    """Rows that are not a multiple of 4 samples long and views whose rows start on 4-byte boundaries only (img[..., 1:1025] of a
    1028-wide tensor): the LDS-DMA engine takes both (tools/dma_probe.hip); in zero mode the right pad is re-zeroed behind the last
    lane of a row's request."""
    g = torch.Generator().manual_seed(5)
    for shape in ((2, 130, 650), (1, 97, 1001), (2, 64, 515), (1, 200, 1278)):
        check(torch.randn(*shape, generator=g), "db4", mode, 3 if shape[1] >= 97 else 2, [_engine.KID_PYRAMID])
        check(torch.randn(*shape, generator=g), "db2", mode, 2, [_engine.KID_PYRAMID])
    for width, off in ((512, 1), (1024, 3), (1022, 2)):
        big = torch.randn(2, 150, width + 4, generator=g)
        view = big[..., off:width + off]
        xd = big.to(dev())[..., off:width + off]
        assert xd.data_ptr() % 16 == 4 * off
        got, kids = run_traced(lambda: ptwt_amd.wavedec2(xd, "db4", mode=mode, level=3))
        assert kids == [_engine.KID_PYRAMID], kids
        want = O.wavedec2(view.numpy().astype(np.float64), "db4", mode=mode, level=3)
        for (n, a), (_, b) in zip(G.flatten_coeffs(got), G.flatten_coeffs(want)):
            assert G.relerr(a.cpu().numpy(), b) < TOL32, (width, off, n)


@pytest.mark.parametrize("wavelet", ["haar", "db2", "db3", "db4"])
def test_pyramid_persistent_units_any_cut(wavelet):
    """Round 5: persistent workgroups.  The rows of the last level of all images, laid end to end, are cut into chunks, a workgroup
    runs the units of its chunk one after the other (end of one image, whole images, start of the next).  MIFWT_OPT_PYR_WGS forces
    chunk counts that make workgroups take several units: same sums in the same order — bit-identical to the one-chunk-per-CU launch
    whatever the cut, and against the oracle."""
    g = torch.Generator().manual_seed(31)
    for shape, level in (((5, 520, 600), 3), ((7, 264, 512), 3), ((3, 333, 517), 2), ((9, 200, 512), 1), ((2, 300, 2048), 3)):
        x = torch.randn(*shape, generator=g)
        for mode in ("reflect", "zero", "symmetric"):
            ref = ptwt_amd.wavedec2(x.to(dev()), wavelet, mode=mode, level=level)
            for wgs in (1, 2, 3, 7):
                _engine.set_option(_engine.OPT_PYR_WGS, wgs)
                try:
                    if wgs == 3:
                        check(x, wavelet, mode, level, [_engine.KID_PYRAMID])
                    got = ptwt_amd.wavedec2(x.to(dev()), wavelet, mode=mode, level=level)
                finally:
                    _engine.set_option(_engine.OPT_PYR_WGS, 0)
                for (n, a), (_, b) in zip(G.flatten_coeffs(got), G.flatten_coeffs(ref)):
                    assert torch.equal(a, b), (shape, wavelet, mode, wgs, n)


@pytest.mark.parametrize("wavelet", ["haar", "db2", "db3", "db4"])
def test_pyramid_tail_wave_and_twelve_wave_form_are_bit_identical_to_the_level_waves(wavelet):
    """Round 5: the last few columns of a level (1024-column planes: 515 = 4 x 128 + 3, 261 = 2 x 128 + 5, 134 = 2 x 64 + 6) go to a
    TAIL wave that filters them in direct form, one lane per (row, column), and workgroups that get by with 4 + 2 + 2 level waves have
    twelve waves.  Same products in the same order: bit-identical to the sixteen-wave form with a level wave for the tail
    (MIFWT_OPT_DEBUG bits 19 / 20), for widths whose tails have every size 1 .. 6 at some level, every mode, 1-3 levels."""
    g = torch.Generator().manual_seed(51)
    for shape, level in (((3, 200, 1024), 3), ((2, 264, 1018), 3), ((2, 136, 1000), 3), ((2, 200, 520), 3), ((2, 300, 1030), 2), ((3, 130, 1032), 1),
                         ((2, 140, 514), 3), ((1, 1024, 1024), 3), ((2, 150, 1010), 3)):
        x = torch.randn(*shape, generator=g).to(dev())
        for mode in MODES:
            got = ptwt_amd.wavedec2(x, wavelet, mode=mode, level=level)
            for dbg in (524288, 1048576):
                _engine.set_option(_engine.OPT_DEBUG, dbg)
                try:
                    ref = ptwt_amd.wavedec2(x, wavelet, mode=mode, level=level)
                finally:
                    _engine.set_option(_engine.OPT_DEBUG, 0)
                for (n, a), (_, b) in zip(G.flatten_coeffs(got), G.flatten_coeffs(ref)):
                    assert torch.equal(a, b), (shape, wavelet, mode, level, dbg, n)
    check(torch.randn(2, 200, 1024, generator=g), wavelet, "symmetric", 3, [_engine.KID_PYRAMID])
    check(torch.randn(2, 136, 1000, generator=g), wavelet, "zero", 3, [_engine.KID_PYRAMID], seg_rows=8)


@pytest.mark.parametrize("wavelet", ["haar", "db2", "db3", "db4", "db5", "sym5"])
def test_pyramid_one_level_periodic_and_ten_taps(wavelet):
    """Round 5: ONE level per launch also with the periodic extension and with ten taps (the reference's own speed-test wavelet and mode,
    examples/speed_tests/timeitconv_2d.py:38-57: db5, periodic) — a single level needs nothing from the far side of the plane but index
    maps.  Every mode, odd / even extents, several row chunks, against the oracle; multi-level calls take the kernel level by level."""
    g = torch.Generator().manual_seed(61)
    ten = wavelet in ("db5", "sym5")
    for shape in ((2, 300, 1000), (3, 131, 517), (2, 96, 1001), (1, 1000, 1000)):
        x = torch.randn(*shape, generator=g, dtype=torch.float32)
        for mode in ALL_MODES:
            if not ten and mode != "periodic":
                continue  # (covered above)
            check(x, wavelet, mode, 1, want_kids=[_engine.KID_PYRAMID])
            check(x, wavelet, mode, 1, want_kids=[_engine.KID_PYRAMID], seg_rows=8)
    x = torch.randn(2, 520, 1000, generator=g, dtype=torch.float32)
    got, kids = run_traced(lambda: ptwt_amd.wavedec2(x.to(dev()), wavelet, mode="periodic", level=3))
    assert kids and kids[0] == _engine.KID_PYRAMID, kids
    check(x, wavelet, "periodic", 3)
    _engine.set_option(_engine.OPT_PYRAMID_MODE, 0)  # auto mode: the reference's shape takes kernel 16 for its first level
    try:
        xr = torch.randn(4, 1000, 1000, generator=g, dtype=torch.float32)
        _, kids = run_traced(lambda: ptwt_amd.wavedec2(xr.to(dev()), "db5", mode="periodic", level=5))
        assert kids[0] == _engine.KID_PYRAMID, kids
        check(xr, "db5", "periodic", 5)
    finally:
        _engine.set_option(_engine.OPT_PYRAMID_MODE, 1)


def test_pyramid_batches_that_do_not_divide_the_chip():
    """Batches around multiples of the CU count / 4 (65, 100 images of 1024^2 took 1.7 / 1.4 x the time per image of 64 until round 4:
    a last round of a few workgroups): chunks that start and end inside images, checked image by image against one-image calls
    (bit-identical: the results do not depend on the cut) and, for the first and last image, against the oracle."""
    g = torch.Generator().manual_seed(32)
    for B, H, W in ((65, 264, 512), (37, 520, 1024), (100, 136, 640)):
        x = torch.randn(B, H, W, generator=g)
        xd = x.to(dev())
        got = ptwt_amd.wavedec2(xd, "db4", level=3)
        for i in (0, 1, B // 2, B - 2, B - 1):
            one = ptwt_amd.wavedec2(xd[i : i + 1], "db4", level=3)
            for (n, a), (_, b) in zip(G.flatten_coeffs(got), G.flatten_coeffs(one)):
                assert torch.equal(a[i : i + 1], b), (B, i, n)
        for i in (0, B - 1):
            want = O.wavedec2(x[i : i + 1].numpy().astype(np.float64), "db4", level=3)
            for (n, a), (_, b) in zip(G.flatten_coeffs(got), G.flatten_coeffs(want)):
                assert G.relerr(a[i : i + 1].cpu().numpy(), b) < TOL32, (B, i, n)


def test_pyramid_randomised_against_per_level_kernels():
    """Random plane shapes (one to four column groups, odd heights, widths that are multiples of 4), batches, filters, modes, level
    counts and row-segment overrides through the multi-level launch against the per-level kernels on the same data (those are
    pinned against the oracle and the goldens); 2e-6 norm-wise per sub-band."""
    rng = np.random.default_rng(77)
    ran = 0
    for trial in range(40):
        wavelet = ["haar", "db2", "db3", "db4"][rng.integers(4)]
        mode = MODES[rng.integers(len(MODES))]
        h = int(rng.integers(40, 400))
        w = 4 * int(rng.integers(10, 1100))
        b = int(rng.integers(1, 5))
        if b * h * w > 3_000_000:
            b, h = 1, min(h, 3_000_000 // w)
        level = int(rng.integers(1, 4))
        seg = int(rng.choice([0, 0, 8, 16]))
        x = torch.randn(b, h, w, device=dev())
        if seg:
            _engine.set_option(_engine.OPT_PAIR_ROWS, seg)
        try:
            got, kids = run_traced(lambda: ptwt_amd.wavedec2(x, wavelet, mode=mode, level=level))
        except RuntimeError:
            continue
        finally:
            _engine.set_option(_engine.OPT_PAIR_ROWS, 0)
        _engine.set_option(_engine.OPT_PYRAMID_MODE, 2)
        _engine.set_option(_engine.OPT_PAIR_MODE, 2)
        try:
            want = ptwt_amd.wavedec2(x, wavelet, mode=mode, level=level)
        finally:
            _engine.set_option(_engine.OPT_PAIR_MODE, 0)
            _engine.set_option(_engine.OPT_PYRAMID_MODE, 1)
        for (n, u), (_, v) in zip(G.flatten_coeffs(got), G.flatten_coeffs(want)):
            assert u.shape == v.shape
            err = float((u - v).norm() / v.norm().clamp_min(1e-30))
            assert err < 2e-6, (trial, wavelet, mode, (b, h, w), level, seg, n, err, kids)
        ran += _engine.KID_PYRAMID in kids
    assert ran >= 20, ran


# ---- the whole pyramid of a small plane in one launch (mifwt_dwt2_fwd_pyramid's second kernel, kernel id 20) ----------------------
ALL_MODES = MODES + ["periodic"]


@pytest.mark.parametrize("mode", ALL_MODES)
@pytest.mark.parametrize("wavelet", ["haar", "db2", "db4", "sym6", "db10"])
def test_small_planes_whole_pyramid_vs_oracle(wavelet, mode):
    """Image patches and small planes: every level in ONE launch, a workgroup per image; all five modes, filters up to 20 taps,
    odd extents, levels deeper than the filter is long (the boundary map folds repeatedly)."""
    _engine.set_option(_engine.OPT_PYRAMID_MODE, 3)  # (wherever it can run: auto leaves few big planes to the per-level kernels)
    g = torch.Generator().manual_seed(41)
    for shape, level in (((5, 64, 64), 3), ((7, 32, 32), None), ((3, 28, 28), 2), ((2, 128, 128), 4), ((4, 37, 53), 3), ((2, 97, 120), None), ((1, 16, 130), 2)):
        x = torch.randn(*shape, generator=g, dtype=torch.float32)
        try:
            want = O.wavedec2(x.numpy().astype(np.float64), wavelet, mode=mode, level=level)
        except (RuntimeError, ValueError):
            continue
        got, kids = run_traced(lambda: ptwt_amd.wavedec2(x.to(dev()), wavelet, mode=mode, level=level))
        if len(want) == 1:  # (level None: the plane is shorter than the filter, nothing to do)
            continue
        if (shape, wavelet) != ((2, 128, 128), "db10"):  # (its two padded LDS images take 182 KB: served level by level)
            assert kids == [_engine.KID_SMALL], (wavelet, mode, shape, kids)
        gf, wf = G.flatten_coeffs(got), G.flatten_coeffs(want)
        assert [n for n, _ in gf] == [n for n, _ in wf]
        for (n, a), (_, b) in zip(gf, wf):
            assert tuple(a.shape) == tuple(np.shape(b)), n
            assert G.relerr(a.cpu().numpy(), b) < TOL32, (wavelet, mode, shape, level, n)


def test_small_planes_views_big_batches_and_deep_levels_of_big_planes():
    _engine.set_option(_engine.OPT_PYRAMID_MODE, 3)
    g = torch.Generator(device=dev()).manual_seed(42)
    # a crop of a wider tensor (row / image strides that are not the extents), channels folded into the batch
    wide = torch.randn(6, 3, 80, 100, device=dev(), generator=g)
    view = wide[:, :, 5:69, 10:74]
    got, kids = run_traced(lambda: ptwt_amd.wavedec2(view, "db3", level=3))
    assert kids == [_engine.KID_SMALL], kids
    want = O.wavedec2(view.cpu().numpy().astype(np.float64), "db3", level=3)
    for (n, a), (_, b) in zip(G.flatten_coeffs(got), G.flatten_coeffs(want)):
        assert G.relerr(a.cpu().numpy(), b) < TOL32, n
    # 20 000 patches: agreement with the per-level kernels on every image
    x = torch.randn(20000, 32, 32, device=dev(), generator=g)
    got, kids = run_traced(lambda: ptwt_amd.wavedec2(x, "db2", level=3))
    assert kids == [_engine.KID_SMALL]
    _engine.set_option(_engine.OPT_PYRAMID_MODE, 2)
    _engine.set_option(_engine.OPT_PAIR_MODE, 2)
    try:
        ref = ptwt_amd.wavedec2(x, "db2", level=3)
    finally:
        _engine.set_option(_engine.OPT_PAIR_MODE, 0)
        _engine.set_option(_engine.OPT_PYRAMID_MODE, 3)
    for (n, a), (_, b) in zip(G.flatten_coeffs(got), G.flatten_coeffs(ref)):
        assert float(((a - b).flatten(1).norm(dim=1) / b.flatten(1).norm(dim=1).clamp_min(1e-30)).max()) < 2e-6, n
    # the deep levels of a bigger pyramid: three levels streamed, the rest in the small-plane launch
    x = torch.randn(4, 1024, 1024, device=dev(), generator=g)
    got, kids = run_traced(lambda: ptwt_amd.wavedec2(x, "db4", level=6))
    assert kids == [_engine.KID_PYRAMID, _engine.KID_SMALL], kids
    want = O.wavedec2(x[:1].cpu().numpy().astype(np.float64), "db4", level=6)
    for (n, a), (_, b) in zip(G.flatten_coeffs(got), G.flatten_coeffs(want)):
        assert G.relerr(a[:1].cpu().numpy(), b) < TOL32, n
    rec = ptwt_amd.waverec2(got, "db4")
    assert float((rec - x).abs().max()) < 2e-5


# ---- the whole reconstruction of a small plane in one launch (mifwt_dwt2_inv_pyramid, kernel id 21) -------------------------------
def _to_dev32(coeffs):
    """Oracle coefficients (fp64 numpy) -> the same container of f32 device tensors, plus their f32 values back as fp64 numpy."""
    def conv(t):
        return torch.from_numpy(np.ascontiguousarray(t)).float().to(dev())
    out, back = [conv(coeffs[0])], []
    for c in coeffs[1:]:
        if isinstance(c, dict):
            out.append({k: conv(v) for k, v in c.items()})
        else:
            out.append(type(c)(*[conv(v) for v in c]) if hasattr(c, "_fields") else tuple(conv(v) for v in c))
    back.append(out[0].cpu().numpy().astype(np.float64))
    for c in out[1:]:
        if isinstance(c, dict):
            back.append({k: v.cpu().numpy().astype(np.float64) for k, v in c.items()})
        else:
            back.append(tuple(v.cpu().numpy().astype(np.float64) for v in c))
    return tuple(out), tuple(back)


@pytest.mark.parametrize("mode", ALL_MODES)
@pytest.mark.parametrize("wavelet", ["haar", "db2", "db4", "sym6", "db10"])
def test_small_planes_whole_reconstruction_vs_oracle(wavelet, mode):
    """waverec2 of image patches and small planes: every level in ONE launch; coefficients of every boundary mode (odd extents: the
    running approximation is trimmed between the levels), filters up to 20 taps."""
    _engine.set_option(_engine.OPT_PYRAMID_MODE, 3)
    rng = np.random.default_rng(43)
    for shape, level in (((5, 64, 64), 3), ((7, 32, 32), None), ((3, 28, 28), 2), ((2, 128, 128), 4), ((4, 37, 53), 3), ((2, 97, 120), None), ((1, 16, 130), 2),
                         ((3, 50, 50), 1)):
        x = rng.standard_normal(shape)
        try:
            coeffs = O.wavedec2(x, wavelet, mode=mode, level=level)
        except (RuntimeError, ValueError):
            continue
        if len(coeffs) == 1:
            continue
        cdev, c32 = _to_dev32(coeffs)
        want = O.waverec2(c32, wavelet)
        got, kids = run_traced(lambda: ptwt_amd.waverec2(cdev, wavelet))
        if (shape, wavelet) not in (((2, 128, 128), "db10"), ((2, 128, 128), "sym6"), ((2, 97, 120), "db10")):  # (more than 160 KB of LDS: level by level)
            assert kids == [_engine.KID_INV_SMALL], (wavelet, mode, shape, kids)
        assert tuple(got.shape) == tuple(want.shape)
        assert G.relerr(got.cpu().numpy(), want) < TOL32, (wavelet, mode, shape, level)


def test_small_planes_reconstruction_separable_big_batches_and_round_trip():
    _engine.set_option(_engine.OPT_PYRAMID_MODE, 3)
    rng = np.random.default_rng(44)
    # separable containers: the running approximation is CROPPED to the next level's detail shape
    x = rng.standard_normal((3, 45, 61))
    coeffs = O.fswavedec2(x, "db3", mode="symmetric", level=3)
    cdev, c32 = _to_dev32(coeffs)
    got, kids = run_traced(lambda: ptwt_amd.fswaverec2(cdev, "db3"))
    assert kids == [_engine.KID_INV_SMALL], kids
    want = O.fswaverec2(c32, "db3")
    assert tuple(got.shape) == tuple(want.shape) and G.relerr(got.cpu().numpy(), want) < TOL32
    # 20 000 patches: round trip, and agreement with the per-level kernels on every image
    g = torch.Generator(device=dev()).manual_seed(45)
    xb = torch.randn(20000, 32, 32, device=dev(), generator=g)
    cb = ptwt_amd.wavedec2(xb, "db2", level=3)
    rec, kids = run_traced(lambda: ptwt_amd.waverec2(cb, "db2"))
    assert kids == [_engine.KID_INV_SMALL], kids
    assert float((rec - xb).abs().max()) < 1e-5
    _engine.set_option(_engine.OPT_PYRAMID_MODE, 2)
    try:
        ref, kids2 = run_traced(lambda: ptwt_amd.waverec2(cb, "db2"))
    finally:
        _engine.set_option(_engine.OPT_PYRAMID_MODE, 3)
    assert _engine.KID_INV_SMALL not in kids2
    assert float(((rec - ref).flatten(1).norm(dim=1) / ref.flatten(1).norm(dim=1)).max()) < 2e-6
    # leading dims and channels folded into the batch; gradients requested: the same launch as a differentiable op (round 4; it used
    # to step aside for the per-level ops)
    cs = ptwt_amd.wavedec2(torch.randn(4, 3, 64, 64, device=dev(), generator=g), "db4", level=2)
    rec, kids = run_traced(lambda: ptwt_amd.waverec2(cs, "db4"))
    assert kids == [_engine.KID_INV_SMALL] and rec.shape == (4, 3, 64, 64)
    cs[0].requires_grad_(True)
    rec2, kids = run_traced(lambda: ptwt_amd.waverec2(cs, "db4"))
    assert kids == [_engine.KID_INV_SMALL] and rec2.requires_grad
    assert torch.equal(rec2.detach(), rec)
    (g0,) = torch.autograd.grad(rec2.square().sum(), cs[0])
    assert g0.shape == cs[0].shape and torch.isfinite(g0).all()


def test_small_plane_kernels_randomised_against_per_level_kernels():
    """Random plane shapes (1 .. 140 samples per axis: planes shorter than the filter, odd extents, one-row planes), batches, filters
    of 2 .. 20 taps, all five modes, 1 .. 6 levels through the one-launch analysis and synthesis kernels (ids 20 / 21) against the
    per-level kernels run in fp64 on the same data (those are pinned against the oracle and the goldens): 1e-6 norm-wise per
    sub-band, or twice the fp32 per-level kernels' own distance from the fp64 result where that is larger (six levels of a 20-tap
    filter on a plane of five rows accumulate more rounding than that on either route)."""
    rng = np.random.default_rng(99)
    g = torch.Generator(device=dev()).manual_seed(99)
    ran_f = ran_i = 0

    def rel(a, r):
        return float((a.double() - r).norm() / r.norm().clamp_min(1e-300))

    for trial in range(80):
        wavelet = ["haar", "db2", "db3", "db4", "sym6", "db8", "db10"][rng.integers(7)]
        mode = ALL_MODES[rng.integers(len(ALL_MODES))]
        h, w = (int(rng.integers(1, 141)) for _ in range(2))
        if trial % 3 == 0:
            h, w = min(h, 40), min(w, 40)
        b = int(rng.integers(1, 40))
        level = int(rng.integers(1, 7))
        x = torch.randn(b, h, w, device=dev(), generator=g)
        _engine.set_option(_engine.OPT_PYRAMID_MODE, 2)
        try:
            ref = ptwt_amd.wavedec2(x.double(), wavelet, mode=mode, level=level)
            lvl = ptwt_amd.wavedec2(x, wavelet, mode=mode, level=level)
            ref_rec = ptwt_amd.waverec2(ref, wavelet)
            lvl_rec = ptwt_amd.waverec2(lvl, wavelet)
            ref_rec32 = ptwt_amd.waverec2([lvl[0].double()] + [type(d)(*(t.double() for t in d)) for d in lvl[1:]], wavelet)
        except (RuntimeError, ValueError):
            continue  # the reference's own refusals (reflect pad >= extent, ...)
        finally:
            _engine.set_option(_engine.OPT_PYRAMID_MODE, 3)
        got, kids = run_traced(lambda: ptwt_amd.wavedec2(x, wavelet, mode=mode, level=level))
        ran_f += kids == [_engine.KID_SMALL]
        for (n, a), (_, r), (_, l32) in zip(G.flatten_coeffs(got), G.flatten_coeffs(ref), G.flatten_coeffs(lvl)):
            assert a.shape == r.shape, (trial, wavelet, mode, (b, h, w), level, n)
            assert rel(a, r) < max(1e-6, 2 * rel(l32, r)), (trial, wavelet, mode, (b, h, w), level, n, kids, rel(a, r), rel(l32, r))
        rec, kids = run_traced(lambda: ptwt_amd.waverec2(lvl, wavelet))  # the same f32 coefficients through both synthesis routes
        ran_i += kids == [_engine.KID_INV_SMALL]
        assert rec.shape == ref_rec.shape == lvl_rec.shape
        assert rel(rec, ref_rec32) < max(1e-6, 2 * rel(lvl_rec, ref_rec32)), (trial, wavelet, mode, (b, h, w), level, kids)
    assert ran_f >= 40 and ran_i >= 40, (ran_f, ran_i)
