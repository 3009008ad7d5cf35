"""The depth-walking fused 3-D analysis kernel (kernel id 24, mifwt_dwt3_fwd_walk.hip) against the fp64 oracle and against the
other 3-D routes (LDS bricks, id 9; composed 2-D planes + depth pass, id 5): every boundary mode — periodic included —, filters
of 2 .. 10 taps, one to three column strips, ragged row groups, odd extents, rows longer than one 1-KiB request, several depth
segments, strided (sliced) inputs.  Reference seam: src/ptwt/conv_transform_3.py:121-141."""
import numpy as np
import pytest
import torch

import ptwt_amd
from ptwt_amd import _engine
from oracle import fwt_oracle as O
from tests import _golden as G
from tests.test_gpu_parity import MODES, TOL32, TOL64, check_tree, dev, to_np

pytestmark = pytest.mark.gpu

WALK = 4  # MIFWT_OPT_TILE_MODE value that routes 3-D analysis levels to the walk kernel wherever it can run


def _walk(fn):
    _engine.set_option(_engine.OPT_TILE_MODE, WALK)
    try:
        _engine.level_events = []
        out = fn()
        kids = [e[1] for e in _engine.level_events]
    finally:
        _engine.level_events = None
        _engine.set_option(_engine.OPT_TILE_MODE, 0)
    return out, kids


@pytest.mark.parametrize("wavelet", ["haar", "db2", "db3", "db4", "db5"])
def test_walk3_vs_oracle(wavelet):
    rng = np.random.default_rng(len(wavelet) + 40)
    flen = len(O.filter_bank(wavelet)[0])
    shapes = [(2, 21, 37, 141), (1, 2 * flen + 1, 2 * flen, 2 * flen + 3), (3, 9, 70, 66), (2, 12, 21, 130), (1, 40, 20, 258),
              (1, 11, 13, 300), (1, 33, 9, 129)]
    for shape in shapes:
        if min(shape[1:]) < flen:  # (the kernel's single-fold boundary map wants every extent >= the filter length)
            continue
        x = rng.standard_normal(shape)
        xg = torch.from_numpy(x).float().to(dev())
        for mode in MODES:
            level = 2 if min(shape[1:]) >= 3 * flen else 1
            try:
                want = O.wavedec3(x, wavelet, mode=mode, level=level)
            except RuntimeError:
                continue
            got, kids = _walk(lambda: ptwt_amd.wavedec3(xg, wavelet, mode=mode, level=level))
            assert kids and kids[0] == 24, (kids, shape, mode)
            check_tree(got, want, TOL32, f"dwt3 walk {wavelet} {mode} {shape}")


@pytest.mark.parametrize("wavelet", ["haar", "db2", "db3", "db4", "db5"])
def test_walk3_f64_vs_oracle(wavelet):
    """The f64 instance of the walk kernel (rows of at most 256 samples = two 1-KiB requests): every mode, 1e-12 against the oracle;
    with MIFWT_OPT_TILE_ROWS = 2 the two-row form of the short filters."""
    rng = np.random.default_rng(len(wavelet) + 140)
    flen = len(O.filter_bank(wavelet)[0])
    shapes = [(2, 21, 37, 141), (1, 2 * flen + 1, 2 * flen, 2 * flen + 3), (3, 9, 70, 66), (2, 12, 21, 130), (1, 40, 20, 256),
              (1, 11, 13, 127), (1, 33, 9, 129)]
    for shape in shapes:
        if min(shape[1:]) < flen:
            continue
        x = rng.standard_normal(shape)
        xg = torch.from_numpy(x).to(dev())
        for mode in MODES:
            level = 2 if min(shape[1:]) >= 3 * flen else 1
            try:
                want = O.wavedec3(x, wavelet, mode=mode, level=level)
            except RuntimeError:
                continue
            for rows in ((0, 2) if flen <= 4 else (0,)):
                _engine.set_option(_engine.OPT_TILE_ROWS, rows)
                try:
                    got, kids = _walk(lambda: ptwt_amd.wavedec3(xg, wavelet, mode=mode, level=level))
                finally:
                    _engine.set_option(_engine.OPT_TILE_ROWS, 0)
                assert kids and kids[0] == 24, (kids, shape, mode)
                assert got[0].dtype == torch.float64
                check_tree(got, want, TOL64, f"dwt3 walk f64 {wavelet} {mode} {shape} rows {rows}")
    # a slice of a bigger tensor (strides larger than the extents, rows that start on odd 8-byte boundaries), short depth segments
    big = torch.from_numpy(rng.standard_normal((2, 40, 45, 141))).to(dev())
    xs = big[:, 3:37, 2:43, 5:138]
    for mode in ("reflect", "periodic"):
        want = O.wavedec3(to_np(xs), wavelet, mode=mode, level=1)
        _engine.set_option(_engine.OPT_ROWS_PER_CHUNK, 8)
        try:
            got, kids = _walk(lambda: ptwt_amd.wavedec3(xs, wavelet, mode=mode, level=1))
        finally:
            _engine.set_option(_engine.OPT_ROWS_PER_CHUNK, 0)
        assert kids == [24], kids
        check_tree(got, want, TOL64, f"dwt3 walk f64 strided {wavelet} {mode}")
        rec, kids = _walk(lambda: ptwt_amd.waverec3(got, wavelet))
        assert G.relerr(to_np(rec[..., :34, :41, :133]), to_np(xs)) < 1e-11, (wavelet, mode)
    # rows of more than 256 doubles are not the kernel's: the composed route serves them
    xg = torch.randn(1, 12, 12, 300, device=dev(), dtype=torch.float64)
    _, kids = _walk(lambda: ptwt_amd.wavedec3(xg, wavelet, mode="zero", level=1))
    assert kids == [5], kids


def test_walk3_many_segments_and_strided_input():
    """Depth segments forced short (8 output slices each), the input a slice of a bigger tensor (strides larger than the extents)."""
    rng = np.random.default_rng(77)
    big = torch.from_numpy(rng.standard_normal((2, 70, 45, 140))).float().to(dev())
    xg = big[:, 3:69, 2:43, 5:137]
    x = to_np(xg).astype(np.float64)
    for wavelet, mode in [("db2", "reflect"), ("db3", "periodic"), ("db2", "zero"), ("db4", "symmetric"), ("haar", "constant")]:
        want = O.wavedec3(x, wavelet, mode=mode, level=1)
        _engine.set_option(_engine.OPT_ROWS_PER_CHUNK, 8)
        try:
            got, kids = _walk(lambda: ptwt_amd.wavedec3(xg, wavelet, mode=mode, level=1))
        finally:
            _engine.set_option(_engine.OPT_ROWS_PER_CHUNK, 0)
        assert kids == [24], kids
        check_tree(got, want, TOL32, f"dwt3 walk segments {wavelet} {mode}")


def test_walk3_agrees_with_bricks_on_config3_shape():
    """One volume of BASELINE configs[2] (256^3, db2, level 3, zero mode): walk kernel vs the brick kernel, band by band."""
    x = torch.randn(1, 256, 256, 256, device=dev())
    _engine.set_option(_engine.OPT_TILE_MODE, 1)
    try:
        ref = ptwt_amd.wavedec3(x, "db2", mode="zero", level=3)
    finally:
        _engine.set_option(_engine.OPT_TILE_MODE, 0)
    got, kids = _walk(lambda: ptwt_amd.wavedec3(x, "db2", mode="zero", level=3))
    assert kids == [24, 24, 24], kids
    for (n, a), (_, b) in zip(G.flatten_coeffs(got), G.flatten_coeffs(ref)):
        assert G.relerr(to_np(a), to_np(b)) < 5e-7, n
    # auto routing: the walk kernel on the 256^3 level, the bricks on the 129^3 / 66^3 levels
    _engine.level_events = []
    try:
        auto = ptwt_amd.wavedec3(x, "db2", mode="zero", level=3)
        kids = [e[1] for e in _engine.level_events]
    finally:
        _engine.level_events = None
    assert kids == [24, 9, 9], kids
    for (n, a), (_, b) in zip(G.flatten_coeffs(auto), G.flatten_coeffs(ref)):
        assert G.relerr(to_np(a), to_np(b)) < 5e-7, n


# ------------------------------------------------------------------ synthesis mirror (kernel id 25, mifwt_dwt3_inv_walk.hip)
@pytest.mark.parametrize("wavelet", ["haar", "db2", "db3", "db4"])
def test_iwalk3_vs_oracle(wavelet):
    """The depth-walking synthesis kernel against the fp64 oracle fed the same f32 coefficients (coefficients of every boundary mode:
    odd extents, i.e. trimmed outputs; one to three column strips; ragged row groups; pieces of 1 .. 5 KiB) and against the bricks.
    Reference seam: src/ptwt/conv_transform_3.py:205-249."""
    rng = np.random.default_rng(len(wavelet) + 51)
    flen = len(O.filter_bank(wavelet)[0])
    for shape in [(2, 21, 37, 141), (1, 2 * flen + 1, 2 * flen, 2 * flen + 3), (3, 9, 70, 66), (1, 12, 21, 260), (2, 40, 33, 128), (1, 20, 9, 300),
                  (1, 9, 10, 400), (2, 10, 12, 180), (2, 9, 9, 40)]:  # (the last three: row pieces of 5 / 2 / 1 KiB)
        if min(shape[1:]) < flen:
            continue
        x = rng.standard_normal(shape)
        for mode in MODES:
            level = 2 if min(shape[1:]) >= 3 * flen else 1
            try:
                coeffs = O.wavedec3(x, wavelet, mode=mode, level=level)
            except RuntimeError:
                continue
            cdev = [torch.from_numpy(coeffs[0]).float().to(dev())] + [{k: torch.from_numpy(v).float().to(dev()) for k, v in c.items()} for c in coeffs[1:]]
            c32 = [cdev[0].cpu().double().numpy()] + [{k: v.cpu().double().numpy() for k, v in c.items()} for c in cdev[1:]]
            want = O.waverec3(c32, wavelet)
            _engine.set_option(_engine.OPT_TILE_MODE, 1)
            try:
                ref = ptwt_amd.waverec3(cdev, wavelet)
            finally:
                _engine.set_option(_engine.OPT_TILE_MODE, 0)
            got, kids = _walk(lambda: ptwt_amd.waverec3(cdev, wavelet))
            assert kids == [25] * level, (wavelet, mode, shape, kids)
            assert tuple(got.shape) == tuple(want.shape)
            assert G.relerr(to_np(got), want) < TOL32, (wavelet, mode, shape)
            assert G.relerr(to_np(got), to_np(ref)) < 5e-7, (wavelet, mode, shape)


@pytest.mark.parametrize("wavelet", ["haar", "db2", "db3", "db4"])
def test_iwalk3_f64_vs_oracle(wavelet):
    """The f64 instance of the synthesis walk (two row pairs per workgroup, pieces of at most 5 KiB): 1e-12 against the oracle, and
    bit-for-bit-close to the composed route."""
    rng = np.random.default_rng(len(wavelet) + 151)
    flen = len(O.filter_bank(wavelet)[0])
    for shape in [(2, 21, 37, 141), (1, 2 * flen + 1, 2 * flen, 2 * flen + 3), (3, 9, 70, 66), (1, 12, 21, 250), (2, 40, 33, 128), (1, 20, 9, 200),
                  (2, 9, 9, 40)]:
        if min(shape[1:]) < flen:
            continue
        x = rng.standard_normal(shape)
        for mode in MODES:
            level = 2 if min(shape[1:]) >= 3 * flen else 1
            try:
                coeffs = O.wavedec3(x, wavelet, mode=mode, level=level)
            except RuntimeError:
                continue
            cdev = [torch.from_numpy(coeffs[0]).to(dev())] + [{k: torch.from_numpy(v).to(dev()) for k, v in c.items()} for c in coeffs[1:]]
            want = O.waverec3(coeffs, wavelet)
            _engine.set_option(_engine.OPT_TILE_MODE, 2)
            try:
                ref = ptwt_amd.waverec3(cdev, wavelet)
            finally:
                _engine.set_option(_engine.OPT_TILE_MODE, 0)
            got, kids = _walk(lambda: ptwt_amd.waverec3(cdev, wavelet))
            assert kids == [25] * level, (wavelet, mode, shape, kids)
            assert got.dtype == torch.float64 and tuple(got.shape) == tuple(want.shape)
            assert G.relerr(to_np(got), want) < TOL64, (wavelet, mode, shape)
            assert G.relerr(to_np(got), to_np(ref)) < TOL64, (wavelet, mode, shape)


def test_walk3_f64_auto_routes_and_round_trip():
    """f64 volumes take the walk kernels from 32^3 (analysis) / 2^16 (synthesis) samples on for four taps (no bricks for f64; thresholds
    per filter length in mifwt_api.hip); round trip to 1e-12."""
    x = torch.randn(2, 128, 96, 160, device=dev(), dtype=torch.float64)
    _engine.level_events = []
    try:
        c = ptwt_amd.wavedec3(x, "db2", mode="symmetric", level=3)
        y = ptwt_amd.waverec3(c, "db2")
        kids = [e[1] for e in _engine.level_events]
    finally:
        _engine.level_events = None
    assert kids == [24, 24, 24, 6, 25, 25], kids
    assert G.relerr(to_np(y[..., :128, :96, :160]), to_np(x)) < TOL64
    want = O.wavedec3(to_np(x), "db2", mode="symmetric", level=3)
    check_tree(c, want, TOL64, "f64 auto route")


def test_iwalk3_many_segments_round_trip_config3_shape():
    """One volume of BASELINE configs[2] (256^3, db2, level 3, zero mode) through both walk kernels: round trip, and the synthesis
    against the bricks; then short depth segments (8 output slice pairs each)."""
    x = torch.randn(1, 256, 256, 256, device=dev())
    c = ptwt_amd.wavedec3(x, "db2", mode="zero", level=3)
    _engine.set_option(_engine.OPT_TILE_MODE, 1)
    try:
        ref = ptwt_amd.waverec3(c, "db2")
    finally:
        _engine.set_option(_engine.OPT_TILE_MODE, 0)
    got, kids = _walk(lambda: ptwt_amd.waverec3(c, "db2"))
    assert kids == [25, 25, 25], kids
    assert (got - x).abs().max().item() < 5e-6
    assert G.relerr(to_np(got), to_np(ref)) < 5e-7
    _engine.set_option(_engine.OPT_ROWS_PER_CHUNK, 8)
    try:
        seg, kids = _walk(lambda: ptwt_amd.waverec3(c, "db2"))
    finally:
        _engine.set_option(_engine.OPT_ROWS_PER_CHUNK, 0)
    assert kids == [25, 25, 25] and torch.equal(seg, got)
    # auto routing: bricks for the 66^3 level, the walk kernel from the 129^3 level on
    _engine.level_events = []
    try:
        auto = ptwt_amd.waverec3(c, "db2")
        kids = [e[1] for e in _engine.level_events]
    finally:
        _engine.level_events = None
    assert kids == [10, 25, 25], kids
    assert G.relerr(to_np(auto), to_np(ref)) < 5e-7


def test_walk3_gradients_agree_with_the_brick_route():
    """A differentiable wavedec3 / waverec3 of a volume big enough for both walk kernels (168 x 160 x 164 > 2^22 samples): the forward runs on kernel 24, the
    analysis adjoint on kernel 25 (zero-mode synthesis launch + border kernel) and vice versa — gradients against the brick route."""
    x = torch.randn(1, 168, 160, 164, device=dev())
    def grads(tile_mode):
        _engine.set_option(_engine.OPT_TILE_MODE, tile_mode)
        try:
            xx = x.clone().requires_grad_(True)
            c = ptwt_amd.wavedec3(xx, "db2", mode="reflect", level=1)
            loss = (c[0] ** 2).sum() + sum((v ** 2).sum() * (i + 2) for i, v in enumerate(c[1].values()))
            gx, = torch.autograd.grad(loss, xx)
            cc = [c[0].detach().clone().requires_grad_(True), {k: v.detach().clone().requires_grad_(True) for k, v in c[1].items()}]
            y = ptwt_amd.waverec3(cc, "db2")
            gc = torch.autograd.grad((y ** 3).sum(), [cc[0], *cc[1].values()])
        finally:
            _engine.set_option(_engine.OPT_TILE_MODE, 0)
        return gx, gc
    _engine.level_events = []
    try:
        gx_w, gc_w = grads(0)
        kids = {(e[0], e[1]) for e in _engine.level_events}
    finally:
        _engine.level_events = None
    assert kids == {("fwd", 24), ("fwd_adj", 25), ("inv", 25), ("inv_adj", 24)}, kids
    gx_b, gc_b = grads(1)
    assert G.relerr(to_np(gx_w), to_np(gx_b)) < 1e-6
    for a, b in zip(gc_w, gc_b):
        assert G.relerr(to_np(a), to_np(b)) < 1e-6


STRIPS = 2097152  # MIFWT_OPT_DEBUG routing bit: kernel 24 keeps its strip form for eight / ten taps


@pytest.mark.parametrize("wavelet", ["db4", "db5"])
def test_walk3_slab_form_vs_oracle_and_strip_form(wavelet):
    """The slab form of kernel 24 (mifwt_dwt3_fwd_slab.hip: eight / ten taps, rows of at most 128 samples — the reference's own 3-D speed
    shape, examples/speed_tests/timeitconv_3d.py): every mode against the oracle, one and several slabs per volume, ragged last slab, odd
    extents, the smallest volume, short depth segments, strided input — and bit-identical to the strip form (the same sums in the same order)."""
    rng = np.random.default_rng(len(wavelet) + 60)
    flen = len(O.filter_bank(wavelet)[0])
    shapes = [(2, 30, 100, 100), (3, 21, 23, 37), (1, 54, 54, 54), (2, 20, 31, 128), (1, flen, flen, flen), (2, 17, 75, 90), (1, 13, 97, 11), (1, 12, 128, 128)]
    for shape in shapes:
        x = rng.standard_normal(shape)
        xg = torch.from_numpy(x).float().to(dev())
        for mode in MODES:
            try:
                want = O.wavedec3(x, wavelet, mode=mode, level=1)
            except RuntimeError:
                continue
            for seg in (0, 2, 5):
                _engine.set_option(_engine.OPT_ROWS_PER_CHUNK, seg)
                try:
                    got, kids = _walk(lambda: ptwt_amd.wavedec3(xg, wavelet, mode=mode, level=1))
                    if seg == 0:
                        _engine.set_option(_engine.OPT_DEBUG, STRIPS)
                        try:
                            strips, kids_s = _walk(lambda: ptwt_amd.wavedec3(xg, wavelet, mode=mode, level=1))
                        finally:
                            _engine.set_option(_engine.OPT_DEBUG, 0)
                finally:
                    _engine.set_option(_engine.OPT_ROWS_PER_CHUNK, 0)
                assert kids == [24], (kids, shape, mode)
                check_tree(got, want, TOL32, f"dwt3 slab {wavelet} {mode} {shape} seg {seg}")
                if seg == 0:
                    assert kids_s == [24]
                    assert torch.equal(got[0], strips[0]) and all(torch.equal(got[1][k], strips[1][k]) for k in got[1]), (shape, mode)
    # a slice of a bigger tensor; the default route of the reference's shape is this kernel for ten taps
    big = torch.from_numpy(rng.standard_normal((2, 40, 45, 120))).float().to(dev())
    xs = big[:, 3:37, 2:43, 5:105]
    for mode in ("reflect", "periodic", "zero"):
        want = O.wavedec3(to_np(xs).astype(np.float64), wavelet, mode=mode, level=2)
        got, kids = _walk(lambda: ptwt_amd.wavedec3(xs, wavelet, mode=mode, level=2))
        assert kids == [24, 24], kids
        check_tree(got, want, TOL32, f"dwt3 slab strided {wavelet} {mode}")
    if wavelet == "db5":  # (where the cost model of the route sends it: 32 volumes here, one volume to the composed route)
        for batch, want_kid in ((32, 24), (1, 5)):
            xg = torch.randn(batch, 100, 100, 100, device=dev())
            _engine.level_events = []
            try:
                ptwt_amd.wavedec3(xg, "db5", mode="periodic", level=1)
                kids = [e[1] for e in _engine.level_events]
            finally:
                _engine.level_events = None
            assert kids == [want_kid], (batch, kids)


def test_slab_form_through_the_public_calls_round_trip_and_separable_face():
    """What a user of the reference's 3-D speed workload sees: `wavedec3` / `fswavedec3` of a batch of ten-tap volumes take the slab form
    by the default route, agree with the oracle, and `waverec3` / `fswaverec3` of their coefficients give the input back."""
    rng = np.random.default_rng(99)
    x = rng.standard_normal((24, 40, 50, 60))
    xg = torch.from_numpy(x).float().to(dev())
    for mode in ("periodic", "reflect"):
        _engine.level_events = []
        try:
            got = ptwt_amd.wavedec3(xg, "db5", mode=mode, level=2)
            fs = ptwt_amd.fswavedec3(xg, "db5", mode=mode, level=1)
            kids = [e[1] for e in _engine.level_events]
        finally:
            _engine.level_events = None
        assert kids[0] == 24 and kids[2] == 24, kids  # (24 volumes of 1.2e5 samples: the slab form's route; the second level: 1.8e4 samples, composed)
        check_tree(got, O.wavedec3(x, "db5", mode=mode, level=2), TOL32, f"wavedec3 db5 {mode}")
        check_tree(fs, O.fswavedec3(x, "db5", mode=mode, level=1), TOL32, f"fswavedec3 db5 {mode}")
        rec = ptwt_amd.waverec3(got, "db5")
        assert G.relerr(to_np(rec[..., :40, :50, :60]), x) < 5e-6, mode
        rec = ptwt_amd.fswaverec3(fs, "db5")
        assert G.relerr(to_np(rec[..., :40, :50, :60]), x) < 5e-6, mode
