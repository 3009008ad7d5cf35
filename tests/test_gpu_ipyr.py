"""GPU parity tests (``-m gpu``) of the streaming multi-level 2-D synthesis launch (``mifwt_dwt2_inv_pyramid``'s second kernel, id 22:
up to three synthesis levels of a big plane per launch, mifwt_dwt2_inv_pyr.hip) against the fp64 numpy oracle
(src/ptwt/conv_transform_2.py:222-249).

Tolerance: fp32 <= 1e-6 norm-wise vs the fp64 oracle on the f32 coefficients.  Every case asserts that the streaming
kernel is the one that ran (``_engine.level_events``), so a silent per-level fallback cannot pass."""
import numpy as np
import pytest
import torch

import ptwt_amd
from oracle import fwt_oracle as O
from ptwt_amd import _engine
from tests import _golden as G

pytestmark = pytest.mark.gpu

TOL32 = 1e-6
MODES = ["reflect", "zero", "constant", "symmetric", "periodic"]


def dev():
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _wherever_it_can_run():
    """Auto mode keeps the kernel to planes of at least 512 columns; most parity cases here are smaller."""
    _engine.set_option(_engine.OPT_PYRAMID_MODE, 1)
    yield
    _engine.set_option(_engine.OPT_PYRAMID_MODE, 0)
    _engine.set_option(_engine.OPT_PAIR_ROWS, 0)


def run_traced(fn):
    _engine.level_events = []
    try:
        out = fn()
        torch.cuda.synchronize()
        kids = [e[1] for e in _engine.level_events]
    finally:
        _engine.level_events = None
    return out, kids


def to_dev32(coeffs):
    """Oracle coefficients (fp64 numpy) -> f32 device tensors in the same container + their f32 values back as fp64 numpy."""
    def conv(t):
        return torch.from_numpy(np.ascontiguousarray(t)).float().to(dev())
    out = [conv(coeffs[0])] + [tuple(conv(v) for v in c) for c in coeffs[1:]]
    back = [out[0].cpu().numpy().astype(np.float64)] + [tuple(v.cpu().numpy().astype(np.float64) for v in c) for c in out[1:]]
    return tuple(out), tuple(back)


def check(shape, wavelet, mode, level, want_kids, seg_rows=0, random_coeffs=False, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(shape)
    c64 = O.wavedec2(x, wavelet, mode=mode, level=level)
    if random_coeffs:  # not the image of an analysis: the synthesis alone
        c64 = (rng.standard_normal(c64[0].shape),) + tuple(tuple(rng.standard_normal(b.shape) for b in lv) for lv in c64[1:])
    cdev, c32 = to_dev32(c64)
    if seg_rows:
        _engine.set_option(_engine.OPT_PAIR_ROWS, seg_rows)
    try:
        got, kids = run_traced(lambda: ptwt_amd.waverec2(cdev, wavelet))
    finally:
        _engine.set_option(_engine.OPT_PAIR_ROWS, 0)
    want = O.waverec2(c32, wavelet)
    assert tuple(got.shape) == tuple(want.shape), (got.shape, want.shape)
    if want_kids is not None:
        assert kids == want_kids, f"launches {kids}, expected {want_kids}"
    err = G.relerr(got.cpu().numpy(), want)
    assert err < TOL32, f"{wavelet} {mode} L{level} {shape}: rel err {err:.3e}"
    return err


K = _engine.KID_INV_PYRAMID if hasattr(_engine, "KID_INV_PYRAMID") else 22


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("wavelet", ["haar", "db2", "db3", "db4", "db5"])
def test_three_levels_vs_oracle(wavelet, mode):
    check((2, 300, 520), wavelet, mode, 3, [K])
    check((1, 203, 333), wavelet, mode, 3, [K], random_coeffs=True, seed=1)  # odd extents: trims between the levels


@pytest.mark.parametrize("level", [1, 2])
@pytest.mark.parametrize("wavelet", ["db2", "db4", "sym5"])
def test_one_and_two_levels(wavelet, level):
    check((3, 264, 520), wavelet, "reflect", level, [K])
    check((2, 131, 259), wavelet, "symmetric", level, [K], random_coeffs=True, seed=2)


@pytest.mark.parametrize("wavelet", ["haar", "db3", "db4", "db5"])
def test_row_segments(wavelet):
    for seg in (32, 40, 64, 104, 1000):
        check((2, 300, 300), wavelet, "reflect", 3, [K], seg_rows=seg, seed=seg)


def test_ragged_and_wide_planes():
    for shape in ((2, 96, 257), (1, 130, 1021), (2, 64, 1280), (1, 80, 1500), (3, 33, 518)):
        check(shape, "db4", "reflect", 3 if shape[1] >= 96 else 2, [K], seed=shape[2])


@pytest.mark.parametrize("wavelet", ["db5", "sym5", "bior3.3"])
def test_ten_taps_on_big_planes(wavelet):
    """The ten-tap path of the streaming synthesis kernel (one row's windows at a time, 40 accumulator registers) beyond the small
    planes of `test_three_levels_vs_oracle`: the reference's own 2-D speed-test geometry (32 x 1000^2 db5 level 5 periodic,
    examples/speed_tests/timeitconv_2d.py:38-57: here 4 images against the numpy oracle), wide and ragged planes, several row segments,
    random coefficient sets with odd extents (trims between the levels)."""
    flen = len(O.filter_bank(wavelet)[0])
    if flen > 10:
        pytest.skip("more than ten taps: not this kernel's path")
    check((4, 1000, 1000), wavelet, "periodic", 5, None, seed=3)
    rng = np.random.default_rng(8)
    c64 = O.wavedec2(rng.standard_normal((4, 1000, 1000)), wavelet, mode="periodic", level=5)
    cdev, c32 = to_dev32(c64)
    got, kids = run_traced(lambda: ptwt_amd.waverec2(cdev, wavelet))
    assert kids[-1] == K, kids
    for shape, level in (((2, 700, 1500), 3), ((1, 1031, 1277), 3), ((2, 520, 1024), 2)):
        check(shape, wavelet, "symmetric", level, None, random_coeffs=True, seed=shape[1])
    for seg in (40, 104):
        check((2, 600, 800), wavelet, "reflect", 3, [K], seg_rows=seg, seed=seg)


def test_more_levels_than_the_launch_takes():
    """Five levels: the two coarsest go first (one two-level launch or the per-level kernels), the three finest in the streaming launch."""
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 512, 640))
    c64 = O.wavedec2(x, "db2", mode="reflect", level=5)
    cdev, c32 = to_dev32(c64)
    got, kids = run_traced(lambda: ptwt_amd.waverec2(cdev, "db2"))
    assert kids[-1] == K and K not in kids[:-1], kids
    assert G.relerr(got.cpu().numpy(), O.waverec2(c32, "db2")) < TOL32


def test_views_of_level_buffers_and_separable_container():
    """The engine's own analysis output (bands = planes of one level buffer, not dense tensors) and the separable container."""
    g = torch.Generator(device=dev()).manual_seed(3)
    x = torch.randn(3, 2, 256, 512, device=dev(), generator=g)
    cs = ptwt_amd.wavedec2(x, "db4", level=3)
    rec, kids = run_traced(lambda: ptwt_amd.waverec2(cs, "db4"))
    assert kids == [K], kids
    assert float((rec - x).abs().max()) < 2e-5
    want = O.waverec2(tuple([cs[0].cpu().double().numpy()] + [tuple(t.cpu().double().numpy() for t in c) for c in cs[1:]]), "db4")
    assert G.relerr(rec.cpu().numpy(), want) < TOL32
    fs = ptwt_amd.fswavedec2(x, "db3", level=3)
    rec, kids = run_traced(lambda: ptwt_amd.fswaverec2(fs, "db3"))
    assert K in kids, kids
    assert float((rec - x).abs().max()) < 2e-5


def test_config2_full_size_round_trip_and_oracle_images():
    """BASELINE config 2's coefficients (64 x 1024 x 1024, db4, level 3) through ONE launch: round trip on all 64 images, the oracle on
    four of them."""
    _engine.set_option(_engine.OPT_PYRAMID_MODE, 0)  # auto mode must pick the kernel here
    g = torch.Generator(device=dev()).manual_seed(11)
    x = torch.randn(64, 1024, 1024, device=dev(), generator=g)
    cs = ptwt_amd.wavedec2(x, "db4", level=3)
    rec, kids = run_traced(lambda: ptwt_amd.waverec2(cs, "db4"))
    assert kids == [K], kids
    assert float((rec - x).abs().max()) < 2e-5
    for i in (0, 21, 42, 63):
        c64 = tuple([cs[0][i].cpu().double().numpy()] + [tuple(t[i].cpu().double().numpy() for t in c) for c in cs[1:]])
        assert G.relerr(rec[i].cpu().numpy(), O.waverec2(c64, "db4")) < TOL32, i
    # linearity on the full batch
    cs2 = ptwt_amd.wavedec2(torch.randn(64, 1024, 1024, device=dev(), generator=g), "db4", level=3)
    mix = tuple([2.0 * cs[0] - 0.5 * cs2[0]] + [tuple(2.0 * a - 0.5 * b for a, b in zip(u, v)) for u, v in zip(cs[1:], cs2[1:])])
    rec2 = ptwt_amd.waverec2(cs2, "db4")
    recm = ptwt_amd.waverec2(mix, "db4")
    assert float(((2.0 * rec - 0.5 * rec2) - recm).norm() / recm.norm()) < 2e-6


def test_randomised_against_per_level_kernels():
    """Random geometries / wavelets / segment lengths against the per-level kernels (bookkeeping net, not oracle evidence)."""
    rng = np.random.default_rng(2024)
    for trial in range(40):
        wavelet = ["haar", "db2", "db3", "db4", "sym3", "sym4", "db5", "sym5"][int(rng.integers(8))]
        level = int(rng.integers(1, 4))
        h, w = int(rng.integers(120, 400)), int(rng.integers(120, 700))  # (planes the one-launch small-plane kernel does not take)
        b = int(rng.integers(1, 4))
        mode = MODES[int(rng.integers(len(MODES)))]
        x = torch.from_numpy(rng.standard_normal((b, h, w))).float().to(dev())
        try:
            cs = ptwt_amd.wavedec2(x, wavelet, mode=mode, level=level)
        except (RuntimeError, ValueError):
            continue  # (pad longer than the plane at a deep level)
        seg = int(rng.choice([0, 32, 48, 96]))
        _engine.set_option(_engine.OPT_PAIR_ROWS, seg)
        got, kids = run_traced(lambda: ptwt_amd.waverec2(cs, wavelet))
        _engine.set_option(_engine.OPT_PAIR_ROWS, 0)
        assert kids in ([K], [_engine.KID_INV_SMALL]), (trial, kids, wavelet, level, h, w)
        _engine.set_option(_engine.OPT_PYRAMID_MODE, 2)
        _engine.set_option(_engine.OPT_PAIR_MODE, 2)
        try:
            ref = ptwt_amd.waverec2(cs, wavelet)
        finally:
            _engine.set_option(_engine.OPT_PAIR_MODE, 0)
            _engine.set_option(_engine.OPT_PYRAMID_MODE, 1)
        assert float((got - ref).norm() / ref.norm()) < 2e-6, (trial, wavelet, level, mode, h, w, seg)
