"""GPU parity tests (``-m gpu``): the HIP path, called through the C ABI by the ptwt_amd host layer, against
the CPU oracle and the committed golden vectors.

Tolerances (norm-wise per sub-band): fp64 <= 1e-12 vs pywt/reference goldens; fp32 <= 1e-6 vs
the reference's fp32 CPU output and vs the fp64 oracle; round trips fp32 <= 1e-6 / fp64 <= 1e-13.
"""
import numpy as np
import pytest
import torch

import ptwt_amd
from oracle import fwt_oracle as O
from ptwt_amd import _engine
from tests import _golden as G

pytestmark = pytest.mark.gpu

TOL64 = 1e-12
TOL32 = 1e-6
MODES = ["reflect", "zero", "constant", "periodic", "symmetric"]


def dev():
    return torch.device("cuda:0")


def to_np(t):
    return t.detach().cpu().numpy()


def check_tree(got, want, tol, what=""):
    gf, wf = G.flatten_coeffs(got), G.flatten_coeffs(want)
    assert [n for n, _ in gf] == [n for n, _ in wf]
    for (n, a), (_, b) in zip(gf, wf):
        assert tuple(a.shape) == tuple(np.shape(b)), (what, n)
        err = G.relerr(to_np(a), b)
        assert err < tol, f"{what} {n}: rel err {err:.3e} >= {tol}"


def test_extension_loaded_and_fast_path_selected():
    lib = _engine.load_library()
    assert lib.mifwt_abi_version() == _engine.ABI_VERSION == 3
    assert _engine.kernel_id(2, torch.float32, "reflect", 8, 64, (1024, 1024)) == 16    # one level through the streaming multi-level kernel (round 4)
    assert _engine.kernel_id(2, torch.float32, "reflect", 8, 64, (640, 640)) == 7       # fused LDS-tile kernel
    assert _engine.kernel_id(2, torch.float32, "reflect", 16, 64, (4096, 4096)) == 1   # fused streaming kernel


def test_kat_ripples_haar_gpu():
    class Haar:
        filter_bank = ([0.5, 0.5], [-0.5, 0.5], [0.5, 0.5], [0.5, -0.5])

        def __len__(self):
            return 2

    x = torch.tensor([56.0, 40.0, 8.0, 24.0, 48.0, 48.0, 40.0, 16.0], device=dev())
    c = ptwt_amd.wavedec(x, Haar(), level=3)
    assert c[0].item() == 35.0 and c[1].item() == -3.0
    assert c[2].tolist() == [16.0, 10.0] and c[3].tolist() == [8.0, -8.0, 0.0, 12.0]


def test_baseline_config1_haar_4096_fp64():
    """BASELINE configs[0]: 1-D Haar, N = 4096, batch 1, fp64, 12 levels, against the reference golden."""
    z, idx = G.load("ptwt_ref.npz")
    case = idx[0]
    assert case["shape"] == [1, 4096] and case["wavelet"] == "haar"
    x = torch.from_numpy(z["r000_x"]).to(dev())
    c = ptwt_amd.wavedec(x, "haar")
    assert [t.shape[-1] for t in c] == [1, 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048]
    for name, t in G.flatten_coeffs(c):
        assert G.relerr(to_np(t), z["r000_" + name]) < TOL64
    rec = ptwt_amd.waverec(c, "haar")
    assert (rec - x).abs().max().item() < 1e-13


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_vs_pywt_goldens_1d(dtype):
    z, idx = G.load("pywt_wavedec1d.npz")
    tol = TOL64 if dtype == torch.float64 else TOL32
    for case in idx:
        x = torch.from_numpy(z[case["key"] + "_x"]).to(dtype).to(dev())
        got = ptwt_amd.wavedec(x, case["wavelet"], mode=case["mode"], level=case["level"])
        assert len(got) == case["ncoef"]
        for i, g in enumerate(got):
            err = G.relerr(to_np(g), z["%s_%d" % (case["key"], i)])
            assert err < tol, (case, i, err)
        rec = ptwt_amd.waverec(got, case["wavelet"])
        assert G.relerr(to_np(rec[..., : x.shape[-1]]), to_np(x)) < (1e-9 if dtype == torch.float64 else 2e-6)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_vs_pywt_goldens_2d(dtype):
    z, idx = G.load("pywt_wavedec2d.npz")
    tol = TOL64 if dtype == torch.float64 else TOL32
    for case in idx:
        k = case["key"]
        x = torch.from_numpy(z[k + "_x"]).to(dtype).to(dev())
        got = ptwt_amd.wavedec2(x, case["wavelet"], mode=case["mode"], level=case["level"])
        assert G.relerr(to_np(got[0]), z[k + "_a"]) < tol, case
        for i, det in enumerate(got[1:]):
            for n, t in zip("hvd", det):
                err = G.relerr(to_np(t), z["%s_%d_%s" % (k, i, n)])
                assert err < tol, (case, i, n, err)
        rec = ptwt_amd.waverec2(got, case["wavelet"])
        assert G.relerr(to_np(rec[..., : x.shape[-2], : x.shape[-1]]), to_np(x)) < (1e-9 if dtype == torch.float64 else 2e-6)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_vs_pywt_goldens_3d(dtype):
    z, idx = G.load("pywt_wavedec3d.npz")
    tol = TOL64 if dtype == torch.float64 else TOL32
    for case in idx:
        k = case["key"]
        x = torch.from_numpy(z[k + "_x"]).to(dtype).to(dev())
        got = ptwt_amd.wavedec3(x, case["wavelet"], mode=case["mode"], level=case["level"])
        assert G.relerr(to_np(got[0]), z[k + "_a"]) < tol, case
        for i, dct in enumerate(got[1:]):
            for key, t in dct.items():
                assert G.relerr(to_np(t), z["%s_%d_%s" % (k, i, key)]) < tol, (case, i, key)
        rec = ptwt_amd.waverec3(got, case["wavelet"])
        s = x.shape
        assert G.relerr(to_np(rec[..., : s[-3], : s[-2], : s[-1]]), to_np(x)) < (1e-9 if dtype == torch.float64 else 2e-6)


def test_vs_reference_goldens_all_entry_points():
    """Every reference golden case (all ten functions, axes, folded batches, f32 + f64) on the HIP path."""
    z, idx = G.load("ptwt_ref.npz")
    for case in idx:
        k = case["key"]
        x = torch.from_numpy(z[k + "_x"]).to(dev())
        kw = {a: (tuple(v) if isinstance(v, list) else v) for a, v in case["kw"].items()}
        coeffs = getattr(ptwt_amd, case["fn"])(x, case["wavelet"], **kw)
        flat = G.flatten_coeffs(coeffs)
        assert [n for n, _ in flat] == case["names"], case
        tol = TOL64 if case["dtype"] == "float64" else TOL32
        for name, val in flat:
            want = z["%s_%s" % (k, name)]
            assert val.dtype == x.dtype and val.device == x.device
            assert tuple(val.shape) == want.shape, (case, name)
            err = G.relerr(to_np(val), want)
            assert err < tol, (case, name, err)
        rkw = {a: v for a, v in kw.items() if a in ("axis", "axes")}
        rec = getattr(ptwt_amd, case["rec"])(coeffs, case["wavelet"], **rkw)
        want = z[k + "_rec"]
        assert tuple(rec.shape) == want.shape, case
        assert G.relerr(to_np(rec), want) < (1e-11 if case["dtype"] == "float64" else 2e-6), case


FUSED_WAVELETS = ["haar", "db2", "db3", "db4", "db5", "db6", "db7", "db8"]  # L = 2..16


@pytest.mark.parametrize("wavelet", FUSED_WAVELETS)
@pytest.mark.parametrize("mode", MODES)
def test_fused_dwt2_vs_oracle(wavelet, mode):
    """The fused streaming kernel (kernel id 1) against the fp64 oracle: several strips, edge strips on both
    sides, odd and even extents, odd row pitches from level 2 on, a chunk boundary inside the image."""
    rng = np.random.default_rng(hash((wavelet, mode)) % (2**32))
    _engine.set_option(5, 2)  # tile mode 2: always the streaming kernel
    try:
        for shape in [(3, 70, 530), (2, 131, 257), (1, 300, 1101), (2, 40, 36)]:
            flen = len(O.filter_bank(wavelet)[0])
            if _engine.kernel_id(2, torch.float32, mode, flen, shape[0], shape[1:]) != 1:
                pytest.fail("fused streaming path not selected")
            x = rng.standard_normal(shape)
            level = 3 if min(shape[1:]) > 4 * flen else 1
            try:
                want = O.wavedec2(x, wavelet, mode=mode, level=level)
            except RuntimeError:
                with pytest.raises(RuntimeError):
                    ptwt_amd.wavedec2(torch.from_numpy(x).float().to(dev()), wavelet, mode=mode, level=level)
                continue
            got = ptwt_amd.wavedec2(torch.from_numpy(x).float().to(dev()), wavelet, mode=mode, level=level)
            check_tree(got, want, TOL32, f"{wavelet} {mode} {shape}")
    finally:
        _engine.set_option(5, 0)


def test_fused_equals_generic():
    """Fused kernel vs the generic axis passes on the same input: differences are fp32 rounding only."""
    x = torch.randn(4, 203, 610, device=dev())
    for mode in MODES:
        fused = ptwt_amd.wavedec2(x, "db4", mode=mode, level=2)
        _engine.set_option(_engine.OPT_FORCE_GENERIC, 1)
        try:
            generic = ptwt_amd.wavedec2(x, "db4", mode=mode, level=2)
        finally:
            _engine.set_option(_engine.OPT_FORCE_GENERIC, 0)
        for (n, a), (_, b) in zip(G.flatten_coeffs(fused), G.flatten_coeffs(generic)):
            assert G.relerr(to_np(a), to_np(b)) < 5e-7, (mode, n)


@pytest.mark.parametrize("wavelet", FUSED_WAVELETS)
def test_fused_idwt2_vs_oracle(wavelet):
    """The fused streaming synthesis kernel (kernel id 2) against the fp64 oracle, incl. odd extents (end trim),
    several strips / chunks and coefficient tensors that are views of the analysis buffers."""
    rng = np.random.default_rng(len(wavelet) * 7 + 1)
    flen = len(O.filter_bank(wavelet)[0])
    _engine.set_option(5, 2)  # tile mode 2: the streaming kernels
    try:
        assert _engine.kernel_id(2, torch.float32, "zero", flen, 4, (300, 1101), direction=1) == 2
        _check_idwt2(wavelet, rng, flen)
    finally:
        _engine.set_option(5, 0)


@pytest.mark.parametrize("wavelet", FUSED_WAVELETS + ["db10", "sym16"])
@pytest.mark.parametrize("tile_rows", [0, 8, 16, 24, 32])
def test_tile_idwt2_vs_oracle(wavelet, tile_rows):
    """The LDS-tile synthesis kernel (kernel id 8, forced) against the fp64 oracle."""
    rng = np.random.default_rng(len(wavelet) * 7 + tile_rows)
    flen = len(O.filter_bank(wavelet)[0])
    _engine.set_option(5, 1)
    _engine.set_option(6, tile_rows)
    try:
        assert _engine.kernel_id(2, torch.float32, "zero", flen, 4, (300, 1101), direction=1) == 8
        _check_idwt2(wavelet, rng, flen)
    finally:
        _engine.set_option(5, 0)
        _engine.set_option(6, 0)


def _check_idwt2(wavelet, rng, flen):
    for shape in [(3, 70, 530), (2, 131, 257), (1, 300, 1101), (2, 40, 36)]:
        x = rng.standard_normal(shape)
        level = 3 if min(shape[1:]) > 4 * flen else 1
        coeffs64 = O.wavedec2(x, wavelet, mode="symmetric", level=level)
        want = O.waverec2(coeffs64, wavelet)
        cgpu = (torch.from_numpy(coeffs64[0]).float().to(dev()),) + tuple(
            tuple(torch.from_numpy(t).float().to(dev()) for t in det) for det in coeffs64[1:])
        got = ptwt_amd.waverec2(cgpu, wavelet)
        assert tuple(got.shape) == want.shape
        assert G.relerr(to_np(got), want) < TOL32, (wavelet, shape)
        # and straight from the analysis output (strided views of the level buffers)
        xg = torch.from_numpy(x).float().to(dev())
        rec = ptwt_amd.waverec2(ptwt_amd.wavedec2(xg, wavelet, mode="symmetric", level=level), wavelet)
        assert G.relerr(to_np(rec[..., : shape[1], : shape[2]]), x) < 2e-6, (wavelet, shape)


def test_fused_idwt2_equals_generic():
    x = torch.randn(4, 203, 610, device=dev())
    c = ptwt_amd.wavedec2(x, "db4", level=2)
    fused = ptwt_amd.waverec2(c, "db4")
    _engine.set_option(_engine.OPT_FORCE_GENERIC, 1)
    try:
        generic = ptwt_amd.waverec2(c, "db4")
    finally:
        _engine.set_option(_engine.OPT_FORCE_GENERIC, 0)
    assert fused.shape == generic.shape
    assert G.relerr(to_np(fused), to_np(generic)) < 5e-7


def test_strided_inputs_and_axes():
    """Non-contiguous inputs / non-default axes go through the stride-aware descriptor, no hidden copies needed."""
    base = torch.randn(3, 50, 2, 66, device=dev(), dtype=torch.float64)
    want = O.wavedec2(to_np(base), "db3", level=2, axes=(1, 3))
    got = ptwt_amd.wavedec2(base, "db3", level=2, axes=(1, 3))
    check_tree(got, want, TOL64, "axes=(1,3)")
    view = torch.randn(2, 64, 96, device=dev())[:, ::2, 8:72]  # strided rows, offset columns
    want = O.wavedec2(to_np(view).astype(np.float64), "db2", level=2)
    check_tree(ptwt_amd.wavedec2(view, "db2", level=2), want, TOL32, "strided view")
    rec = ptwt_amd.waverec2(ptwt_amd.wavedec2(view, "db2", level=2), "db2")
    assert (rec - view).abs().max().item() < 5e-6


@pytest.mark.parametrize("ndim,fn,rec", [(1, "wavedec", "waverec"), (2, "wavedec2", "waverec2"), (3, "wavedec3", "waverec3"),
                                         (2, "fswavedec2", "fswaverec2"), (3, "fswavedec3", "fswaverec3")])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_roundtrip_random(ndim, fn, rec, dtype):
    shape = {1: (5, 1001), 2: (3, 65, 130), 3: (2, 33, 34, 35)}[ndim]
    x = torch.randn(*shape, device=dev(), dtype=dtype)
    for wavelet in ("haar", "db4", "sym5"):
        c = getattr(ptwt_amd, fn)(x, wavelet, mode="symmetric", level=2)
        y = getattr(ptwt_amd, rec)(c, wavelet)
        sl = tuple(slice(0, s) for s in x.shape)
        err = G.relerr(to_np(y[sl]), to_np(x))
        assert err < (1e-6 if dtype == torch.float32 else 5e-12), (fn, wavelet, err)  # sym5 taps are PR-exact to ~1e-13 only


def test_long_filters_generic_path():
    """sym16 (32 taps) and coif17 (102 taps): outside the fused envelope, served by the generic kernels."""
    x = torch.randn(2, 150, 170, device=dev(), dtype=torch.float64)
    for wavelet in ("sym16", "coif17"):
        want = O.wavedec2(to_np(x), wavelet, mode="symmetric", level=1)
        got = ptwt_amd.wavedec2(x, wavelet, mode="symmetric", level=1)
        check_tree(got, want, TOL64, wavelet)


def test_full_size_config2_properties():
    """BASELINE configs[1] at full size (64 x 1024 x 1024 fp32, db4, level 3): oracle check on two images,
    linearity and the round trip."""
    torch.manual_seed(0)
    x = torch.randn(64, 1024, 1024, device=dev())
    c = ptwt_amd.wavedec2(x, "db4", level=3)
    assert [tuple(t.shape[-2:]) for t in (c[0], c[1][0], c[2][0], c[3][0])] == [(134, 134), (134, 134), (261, 261), (515, 515)]
    for b in (0, 63):
        want = O.wavedec2(to_np(x[b]).astype(np.float64), "db4", level=3)
        got = tuple([c[0][b]] + [tuple(t[b] for t in det) for det in c[1:]])
        check_tree(got, want, TOL32, f"image {b}")
    # linearity: T(2x + y) = 2 T(x) + T(y)
    y = torch.randn_like(x)
    cy = ptwt_amd.wavedec2(y, "db4", level=3)
    cz = ptwt_amd.wavedec2(2 * x + y, "db4", level=3)
    for (n, a), (_, b), (_, d) in zip(G.flatten_coeffs(cz), G.flatten_coeffs(c), G.flatten_coeffs(cy)):
        assert G.relerr(to_np(a[:4]), to_np((2 * b + d)[:4])) < 2e-6, n
    rec = ptwt_amd.waverec2(c, "db4")
    assert rec.shape == x.shape
    assert (rec - x).abs().max().item() < 5e-6
    assert G.relerr(to_np(rec[:2]), to_np(x[:2])) < 1e-6


def test_config3_wavedec3_db2():
    """BASELINE configs[2] shape family (wavedec3 db2 level 3, mode zero) at a reduced batch vs the oracle."""
    x = torch.randn(1, 128, 128, 128, device=dev())
    want = O.wavedec3(to_np(x).astype(np.float64), "db2", level=3)
    got = ptwt_amd.wavedec3(x, "db2", level=3)
    check_tree(got, want, TOL32, "wavedec3")
    rec = ptwt_amd.waverec3(got, "db2")
    assert (rec - x).abs().max().item() < 5e-6


# ------------------------------------------------------------------ streaming single-axis routes (kernel ids 3-6)
def test_stream_routes_selected():
    kid = _engine.kernel_id
    assert kid(1, torch.float32, "reflect", 8, 4, (1000,)) == 3 and kid(1, torch.float32, "zero", 8, 4, (1000,), direction=1) == 4
    assert kid(1, torch.float64, "reflect", 2, 1, (4096,)) == 3
    assert kid(3, torch.float32, "zero", 4, 8, (256, 256, 256)) == 24 and kid(3, torch.float32, "zero", 4, 4, (129, 129, 129)) == 9  # 3-D analysis: depth-walking kernel on big volumes (round 4), LDS bricks below
    assert kid(3, torch.float32, "zero", 4, 8, (129, 129, 129)) == 24 and kid(3, torch.float32, "zero", 4, 8, (66, 66, 66)) == 9  # (round 6: from 2^21 samples on with eight volumes or more — config 3's second level)
    assert kid(3, torch.float32, "periodic", 10, 32, (100, 100, 100)) == 24 and kid(3, torch.float32, "periodic", 10, 2, (100, 100, 100)) == 5  # ten taps: the slab form where the batch fills the chip
    assert kid(3, torch.float32, "zero", 8, 8, (256, 256, 256)) == 24 and kid(3, torch.float32, "zero", 12, 8, (256, 256, 256)) == 5 and kid(3, torch.float32, "zero", 4, 8, (256, 256, 256), direction=1) == 25 and kid(3, torch.float32, "zero", 4, 8, (64, 64, 64), direction=1) == 10 and kid(3, torch.float32, "zero", 10, 8, (256, 256, 256), direction=1) == 6
    # f64 volumes (round 5): the f64 instances of the walk kernels from 32^3 samples on (rows of at most 256 samples; synthesis up to 8 taps),
    # below and beyond the composed route — the f64 tile kernel over every depth slice + one depth pass — instead of three axis passes
    assert kid(3, torch.float64, "zero", 4, 2, (33, 34, 35)) == 24 and kid(3, torch.float64, "zero", 4, 2, (33, 34, 35), direction=1) == 6
    assert kid(3, torch.float64, "zero", 4, 2, (41, 42, 43), direction=1) == 25
    assert kid(3, torch.float64, "zero", 4, 2, (20, 21, 22)) == 5 and kid(3, torch.float64, "zero", 4, 2, (20, 21, 22), direction=1) == 6
    assert kid(3, torch.float64, "zero", 4, 2, (40, 40, 300)) == 5 and kid(3, torch.float64, "zero", 10, 2, (64, 64, 64), direction=1) == 6
    assert kid(3, torch.float64, "zero", 10, 2, (128, 128, 128)) == 5  # ten taps: the composed route stays ahead in f64
    assert kid(3, torch.float64, "zero", 8, 2, (100, 100, 100)) == 24 and kid(3, torch.float64, "zero", 8, 2, (66, 66, 66)) == 5
    assert kid(3, torch.float64, "zero", 6, 2, (66, 66, 66)) == 24 and kid(3, torch.float64, "zero", 6, 2, (66, 66, 66), direction=1) == 6
    assert kid(3, torch.float64, "zero", 8, 2, (100, 100, 100), direction=1) == 25 and kid(3, torch.float64, "zero", 8, 2, (256, 256, 256), direction=1) == 6
    assert kid(3, torch.float64, "zero", 24, 2, (60, 60, 60)) == 3  # f64, long filter: inner pass + two outer passes
    assert kid(3, torch.float32, "zero", 8, 8, (54, 54, 54)) == 5  # 8 taps on a small volume: composed route, not the walking kernel
    assert kid(2, torch.float64, "reflect", 8, 2, (64, 64)) == 7 and kid(2, torch.float64, "reflect", 8, 2, (64, 64), direction=1) == 8  # f64 tiles
    assert kid(2, torch.float64, "reflect", 24, 2, (64, 64)) == 3  # f64, long filter: inner + outer pass
    assert kid(2, torch.float32, "symmetric", 32, 2, (300, 300)) == 7  # sym16 analysis: LDS-tile kernel
    assert kid(2, torch.float32, "symmetric", 32, 2, (300, 300), direction=1) == 8  # sym16 synthesis: LDS tiles
    assert kid(2, torch.float32, "symmetric", 102, 2, (300, 300)) == 0  # coif17: generic passes


STREAM_WAVELETS = ["haar", "db2", "db3", "db4", "db6", "db8", "db9", "db10", "db12", "sym16"]  # L = 2 .. 32


@pytest.mark.parametrize("wavelet", STREAM_WAVELETS)
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_stream_axis_kernels_vs_oracle(wavelet, dtype):
    """1-D (inner-axis kernels), 2-D f64 / long filters (inner + outer pass) and 3-D (f32: fused planes + depth pass,
    f64: inner + two outer passes) against the fp64 oracle, every boundary mode, odd extents, ragged strips."""
    rng = np.random.default_rng(len(wavelet) * 13 + (dtype == torch.float64))
    flen = len(O.filter_bank(wavelet)[0])
    tol = TOL32 if dtype == torch.float32 else TOL64
    rt = 2e-6 if dtype == torch.float32 else 1e-9
    shapes = {1: [(3, 1027), (1, 4 * flen + 1), (5, 2 * flen)], 2: [(2, 67, 3 * flen + 70)], 3: [(2, 2 * flen + 3, 2 * flen + 6, 2 * flen + 9)]}
    fns = {1: ("wavedec", "waverec", O.wavedec), 2: ("wavedec2", "waverec2", O.wavedec2), 3: ("wavedec3", "waverec3", O.wavedec3)}
    for nd, shape_list in shapes.items():
        fn, rec, ofn = fns[nd]
        for shape in shape_list:
            x = rng.standard_normal(shape)
            xg = torch.from_numpy(x).to(dtype).to(dev())
            for mode in MODES:
                try:
                    want = ofn(x, wavelet, mode=mode, level=2 if min(shape[1:]) >= 3 * flen else 1)
                except RuntimeError:
                    continue
                got = getattr(ptwt_amd, fn)(xg, wavelet, mode=mode, level=2 if min(shape[1:]) >= 3 * flen else 1)
                check_tree(got, want, tol, f"{fn} {wavelet} {mode} {shape} {dtype}")
                back = getattr(ptwt_amd, rec)(got, wavelet)
                sl = tuple(slice(0, s) for s in shape)
                assert G.relerr(to_np(back[sl]), x) < rt, (fn, wavelet, mode, shape)


def test_stream_equals_generic():
    for shape, fn, rec in [((4, 1003), "wavedec", "waverec"), ((2, 37, 41, 45), "wavedec3", "waverec3")]:
        x = torch.randn(*shape, device=dev())
        for mode in MODES:
            fast = getattr(ptwt_amd, fn)(x, "db3", mode=mode, level=2)
            yf = getattr(ptwt_amd, rec)(fast, "db3")
            _engine.set_option(_engine.OPT_FORCE_GENERIC, 1)
            try:
                slow = getattr(ptwt_amd, fn)(x, "db3", mode=mode, level=2)
                ys = getattr(ptwt_amd, rec)(slow, "db3")
            finally:
                _engine.set_option(_engine.OPT_FORCE_GENERIC, 0)
            for (n, a), (_, b) in zip(G.flatten_coeffs(fast), G.flatten_coeffs(slow)):
                assert G.relerr(to_np(a), to_np(b)) < 5e-7, (fn, mode, n)
            assert G.relerr(to_np(yf), to_np(ys)) < 5e-7, (fn, mode)


def test_half_storage_extension():
    """float16 storage / float32 arithmetic (C ABI MIFWT_F16; BASELINE configs[4] dtype).  The reference rejects
    float16, so the oracle is the fp64 transform of the fp16-quantised input; tolerance = fp16 output rounding
    (5e-4 norm-wise per sub-band)."""
    x = torch.randn(2, 200, 232).to(torch.float16)
    with pytest.raises(ValueError):
        ptwt_amd.wavedec2(x.to(dev()), "sym16", level=1)
    ptwt_amd.set_half_storage(True)
    try:
        for wavelet, fn, ofn in [("sym16", "fswavedec2", O.fswavedec2), ("db4", "wavedec2", O.wavedec2)]:
            want = ofn(x.double().numpy(), wavelet, mode="symmetric", level=2)
            got = getattr(ptwt_amd, fn)(x.to(dev()), wavelet, mode="symmetric", level=2)
            for (n, a), (_, b) in zip(G.flatten_coeffs(got), G.flatten_coeffs(want)):
                assert a.dtype == torch.float16
                assert G.relerr(to_np(a.double()), b) < 5e-4, (wavelet, n)
        x1 = torch.randn(3, 1001).to(torch.float16)
        c = ptwt_amd.wavedec(x1.to(dev()), "db4", level=2)
        y = ptwt_amd.waverec(c, "db4")
        assert G.relerr(to_np(y[..., :1001].double()), x1.double().numpy()) < 2e-3  # four roundings to f16 on the way
        # ... and step by step at 5e-4: each synthesis level against the oracle fed the same f16 coefficients
        cur = c[0]
        for d in c[1:]:
            cur = cur[..., : d.shape[-1]]
            nxt = ptwt_amd.waverec([cur, d], "db4")
            ref = O.waverec([to_np(cur.double()), to_np(d.double())], "db4")
            assert G.relerr(to_np(nxt.double()), ref) < 5e-4
            cur = nxt
    finally:
        ptwt_amd.set_half_storage(False)


@pytest.mark.parametrize("wavelet", ["db9", "db10", "db12", "sym16"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_tile_dwt2_long_filters_and_half(wavelet, dtype):
    """18 / 20 / 24 / 32-tap filters and f16 storage on the LDS-tile kernel (kernel id 7) vs the fp64 oracle of the
    (quantised) input; f16 tolerance = output rounding, 5e-4 norm-wise."""
    rng = np.random.default_rng(len(wavelet) + 100)
    flen = len(O.filter_bank(wavelet)[0])
    tol = TOL32 if dtype == torch.float32 else 5e-4
    if dtype == torch.float16:
        ptwt_amd.set_half_storage(True)
    _engine.set_option(7, 2)  # MFMA mode 2: the vector tile kernel also for f16 long filters
    try:
        for tr in (0, 8, 24):
            _engine.set_option(6, tr)
            for shape in [(2, 131, 3 * flen + 70), (1, 2 * flen, 2 * flen + 1)]:
                assert _engine.kernel_id(2, dtype, "symmetric", flen, shape[0], shape[1:]) == 7
                xq = torch.from_numpy(rng.standard_normal(shape)).to(dtype)
                for mode in MODES:
                    level = 2 if min(shape[1:]) > 4 * flen else 1
                    try:
                        want = O.wavedec2(xq.double().numpy(), wavelet, mode=mode, level=level)
                    except RuntimeError:
                        continue
                    got = ptwt_amd.wavedec2(xq.to(dev()), wavelet, mode=mode, level=level)
                    for (n, a), (_, b) in zip(G.flatten_coeffs(got), G.flatten_coeffs(want)):
                        assert a.dtype == dtype
                        assert G.relerr(to_np(a.double()), b) < tol, (wavelet, mode, shape, n, tr)
    finally:
        _engine.set_option(6, 0)
        _engine.set_option(7, 0)
        ptwt_amd.set_half_storage(False)


@pytest.mark.parametrize("wavelet", ["db9", "db10", "db12", "db14", "sym16"])
def test_mfma_dwt2_long_filters_half(wavelet):
    """The matrix-core (banded-Toeplitz MFMA) analysis kernel (kernel id 11; f16 storage, 18..32 taps) against the fp64
    oracle of the quantised input: every mode, ragged tiles, odd pitches (2-byte-aligned rows -> per-element staging),
    even pitches (dword staging), several tiles per persistent workgroup; and against the vector tile kernel.
    Tolerance 5e-4 norm-wise (f16 output rounding 2.1e-4 + one f16 rounding of the intermediate image)."""
    rng = np.random.default_rng(len(wavelet) + 200)
    flen = len(O.filter_bank(wavelet)[0])
    ptwt_amd.set_half_storage(True)
    try:
        for shape in [(2, 131, 3 * flen + 70), (1, 2 * flen, 2 * flen + 1), (3, 300, 402), (40, 96, 200)]:
            assert _engine.kernel_id(2, torch.float16, "symmetric", flen, shape[0], shape[1:]) == 11
            xq = torch.from_numpy(rng.standard_normal(shape)).to(torch.float16)
            for mode in MODES:
                level = 2 if min(shape[1:]) > 4 * flen else 1
                try:
                    want = O.wavedec2(xq.double().numpy(), wavelet, mode=mode, level=level)
                except RuntimeError:
                    continue
                n_walk, n_tile = _engine.launch_count(_engine.VARIANT_FWD_MFMA_WALK), _engine.launch_count(_engine.VARIANT_FWD_MFMA_TILE)
                got = ptwt_amd.wavedec2(xq.to(dev()), wavelet, mode=mode, level=level)
                # kernel id 11 is answered by two kernels: the one that walks down column panels must have served every level
                assert _engine.launch_count(_engine.VARIANT_FWD_MFMA_WALK) - n_walk == level, (wavelet, mode, shape)
                assert _engine.launch_count(_engine.VARIANT_FWD_MFMA_TILE) == n_tile
                vec = got
                if flen in (18, 20, 24, 32):  # lengths the vector tile kernel is instantiated for
                    _engine.set_option(7, 2)
                    try:
                        vec = ptwt_amd.wavedec2(xq.to(dev()), wavelet, mode=mode, level=level)
                    finally:
                        _engine.set_option(7, 0)
                # one f16 rounding of the intermediate + one of the output per level: 5e-4 PER LEVEL, each level against the oracle
                # fed the same f16-rounded approximation the kernel was fed (the levels inside the multi-level call are these
                # same launches: bit-identical)
                cur = xq.to(dev())
                for lev in range(1, level + 1):
                    one = ptwt_amd.wavedec2(cur, wavelet, mode=mode, level=1)
                    ref = O.wavedec2(cur.double().cpu().numpy(), wavelet, mode=mode, level=1)
                    for (n, a), (_, b) in zip(G.flatten_coeffs(one), G.flatten_coeffs(ref)):
                        assert a.dtype == torch.float16
                        assert G.relerr(to_np(a.double()), b) < 5e-4, (wavelet, mode, shape, lev, n)
                    for a, b in zip(one[1], got[level + 1 - lev]):
                        assert torch.equal(a, b), (wavelet, mode, shape, lev)
                    cur = one[0]
                assert torch.equal(cur, got[0])
                del want
                for (n, a), (_, c) in zip(G.flatten_coeffs(got), G.flatten_coeffs(vec)):
                    assert G.relerr(to_np(a.double()), to_np(c.double())) < (5e-4 if level == 1 else 1e-3) + 1e-4, (wavelet, mode, shape, n)
    finally:
        _engine.set_option(7, 0)
        ptwt_amd.set_half_storage(False)


@pytest.mark.parametrize("wavelet", ["db9", "db10", "db12", "db14", "sym16"])
def test_mfma_idwt2_long_filters_half(wavelet):
    """The matrix-core synthesis kernel (kernel id 23; f16 storage, 18..32 taps; src/ptwt/conv_transform_2.py:222-249) against the
    fp64 oracle fed the same f16 coefficients: random coefficient sets (not images of an analysis), ragged tiles in both directions, odd
    extents (trims), odd and padded pitches, several tiles per workgroup, segments of any length; and the two-level / separable
    containers.  Tolerance 5e-4 norm-wise per level (f16 output rounding 2.1e-4 + one f16 rounding of the intermediate image)."""
    rng = np.random.default_rng(len(wavelet) + 300)
    flen = len(O.filter_bank(wavelet)[0])
    ptwt_amd.set_half_storage(True)
    _engine.set_option(7, 4)  # wherever it can run (auto: from 512 tiles of 32 x 128 samples on)
    try:
        for shape, seg in [((2, 131, 3 * flen + 70), 0), ((1, 2 * flen, 2 * flen + 1), 0), ((3, 300, 402), 3), ((40, 96, 200), 1), ((1, 700, 1031), 5)]:
            for mode in ("reflect", "periodic", "zero"):
                try:
                    c64 = O.wavedec2(rng.standard_normal(shape), wavelet, mode=mode, level=1)
                except RuntimeError:
                    continue
                cq = [torch.from_numpy(rng.standard_normal(c64[0].shape)).half()] + [tuple(torch.from_numpy(rng.standard_normal(b.shape)).half() for b in c64[1])]
                want = O.waverec2((cq[0].double().numpy(), tuple(t.double().numpy() for t in cq[1])), wavelet)
                cdev = (cq[0].to(dev()), tuple(t.to(dev()) for t in cq[1]))
                _engine.set_option(6, seg)
                _engine.level_events = []
                try:
                    got = ptwt_amd.waverec2(cdev, wavelet)
                    torch.cuda.synchronize()
                    kids = [e[1] for e in _engine.level_events]
                finally:
                    _engine.level_events = None
                    _engine.set_option(6, 0)
                assert kids == [23], kids
                assert got.dtype == torch.float16 and tuple(got.shape) == tuple(want.shape)
                assert G.relerr(to_np(got.double()), want) < 5e-4, (wavelet, mode, shape, seg)
        # views with a padded pitch (what the engine's own analysis returns for these planes) and a two-level separable round trip
        x = torch.from_numpy(rng.standard_normal((2, 500, 620))).half().to(dev())
        cs = ptwt_amd.fswavedec2(x, wavelet, mode="symmetric", level=2)
        assert cs[1]["dd"].stride(-2) != cs[1]["dd"].shape[-1]  # (128-byte aligned rows)
        _engine.level_events = []
        rec = ptwt_amd.fswaverec2(cs, wavelet)
        torch.cuda.synchronize()
        kids = [e[1] for e in _engine.level_events]
        _engine.level_events = None
        assert kids == [23, 23], kids
        want = O.fswaverec2(tuple([cs[0].double().cpu().numpy()] + [{k: v.double().cpu().numpy() for k, v in d.items()} for d in cs[1:]]), wavelet)
        assert G.relerr(to_np(rec.double()), want) < 1e-3  # two levels of f16 storage
        assert G.relerr(to_np(rec.double()[..., :500, :620]), x.double().cpu().numpy()) < 2e-3
    finally:
        _engine.set_option(6, 0)
        _engine.set_option(7, 0)
        _engine.level_events = None
        ptwt_amd.set_half_storage(False)


@pytest.mark.parametrize("wavelet", FUSED_WAVELETS)
@pytest.mark.parametrize("tile_rows", [0, 8, 12, 16, 20, 24])
def test_tile_dwt2_vs_oracle(wavelet, tile_rows):
    """The LDS-tile analysis kernel (kernel id 7, forced) against the fp64 oracle: every mode, ragged tiles in both
    directions, odd extents, strided LL input of the deeper levels, planes smaller than the halo."""
    rng = np.random.default_rng(len(wavelet) * 31 + tile_rows)
    flen = len(O.filter_bank(wavelet)[0])
    _engine.set_option(5, 1)
    _engine.set_option(6, tile_rows)
    try:
        assert _engine.kernel_id(2, torch.float32, "reflect", flen, 2, (131, 257)) == 7
        for shape in [(3, 70, 530), (2, 131, 257), (2, 40, 36), (1, 2 * flen, 2 * flen + 1)]:
            x = rng.standard_normal(shape)
            level = 3 if min(shape[1:]) > 4 * flen else 1
            for mode in MODES:
                try:
                    want = O.wavedec2(x, wavelet, mode=mode, level=level)
                except RuntimeError:
                    continue
                got = ptwt_amd.wavedec2(torch.from_numpy(x).float().to(dev()), wavelet, mode=mode, level=level)
                check_tree(got, want, TOL32, f"tile {wavelet} {mode} {shape}")
    finally:
        _engine.set_option(5, 0)
        _engine.set_option(6, 0)


@pytest.mark.parametrize("wavelet", ["haar", "db2", "db3"])
def test_fused_dwt3_tile_vs_oracle_and_composed(wavelet):
    """The fully fused LDS-brick 3-D analysis kernel (kernel id 9) against the fp64 oracle (all modes, ragged bricks on
    every axis, tiny volumes) and against the composed route (fused 2-D planes + depth pass, tile mode 2)."""
    rng = np.random.default_rng(len(wavelet) + 5)
    flen = len(O.filter_bank(wavelet)[0])
    # (.., 130) / (.., 128): one or two coefficient columns behind a full brick column — the last column tile takes them along
    for shape in [(2, 21, 37, 141), (1, 2 * flen + 1, 2 * flen, 2 * flen + 3), (3, 9, 70, 66), (2, 12, 21, 130), (1, 11, 20, 128)]:
        assert _engine.kernel_id(3, torch.float32, "reflect", flen, shape[0], shape[1:]) == 9
        x = rng.standard_normal(shape)
        xg = torch.from_numpy(x).float().to(dev())
        for mode in MODES:
            level = 2 if min(shape[1:]) >= 3 * flen else 1
            try:
                want = O.wavedec3(x, wavelet, mode=mode, level=level)
            except RuntimeError:
                continue
            got = ptwt_amd.wavedec3(xg, wavelet, mode=mode, level=level)
            check_tree(got, want, TOL32, f"dwt3 tile {wavelet} {mode} {shape}")
            # default = slice-per-wave bricks 2 x 4 x 64; tile-rows option 3: slice-per-wave 3 x 4 x 64, 2 / 1: the row-dealt
            # bricks 2 x 4 x 64 / 4 x 4 x 64 — all compute the same sums in the same order
            for opt6, opt1 in [(1, 0), (2, 0), (3, 0)]:
                _engine.set_option(6, opt6)
                try:
                    other = ptwt_amd.wavedec3(xg, wavelet, mode=mode, level=level)
                finally:
                    _engine.set_option(6, 0)
                check_tree(other, want, TOL32, f"dwt3 tile variant {opt6}/{opt1} {wavelet} {mode}")
                for (n, a), (_, b) in zip(G.flatten_coeffs(got), G.flatten_coeffs(other)):
                    assert torch.equal(a, b), (wavelet, mode, shape, n, opt6, opt1)
            _engine.set_option(5, 2)
            try:
                comp = ptwt_amd.wavedec3(xg, wavelet, mode=mode, level=level)
            finally:
                _engine.set_option(5, 0)
            for (n, a), (_, b) in zip(G.flatten_coeffs(got), G.flatten_coeffs(comp)):
                assert G.relerr(to_np(a), to_np(b)) < 5e-7, (wavelet, mode, shape, n)


@pytest.mark.parametrize("wavelet", ["haar", "db2", "db3", "db4"])
def test_fused_idwt3_tile_vs_oracle_and_composed(wavelet):
    """The fully fused LDS-brick 3-D synthesis kernel (kernel id 10) against the fp64 oracle (coefficients of every boundary mode:
    odd extents, i.e. trimmed outputs; ragged bricks on every axis; tiny volumes; more than one column tile) and against the composed
    route (fused 2-D planes + depth pass, tile mode 2) on the same f32 coefficients."""
    rng = np.random.default_rng(len(wavelet) + 11)
    flen = len(O.filter_bank(wavelet)[0])
    for shape in [(2, 21, 37, 141), (1, 2 * flen + 1, 2 * flen, 2 * flen + 3), (3, 9, 70, 66), (1, 12, 21, 260), (2, 40, 33, 128)]:
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
            _engine.level_events = []
            try:
                got = ptwt_amd.waverec3(cdev, wavelet)
                kids = [e[1] for e in _engine.level_events]
            finally:
                _engine.level_events = None
            assert kids == [10] * level, (wavelet, mode, shape, kids)
            assert tuple(got.shape) == tuple(want.shape)
            assert G.relerr(to_np(got), want) < TOL32, (wavelet, mode, shape)
            _engine.set_option(5, 2)
            try:
                comp = ptwt_amd.waverec3(cdev, wavelet)
            finally:
                _engine.set_option(5, 0)
            assert G.relerr(to_np(got), to_np(comp)) < 5e-7, (wavelet, mode, shape)
    # separable containers (the running approximation is cropped to the detail shape) and coefficient views with foreign strides
    x = torch.randn(2, 23, 30, 45, device=dev())
    c = ptwt_amd.fswavedec3(x, wavelet, level=2)
    rec = ptwt_amd.fswaverec3(c, wavelet)
    assert (rec[..., :23, :30, :45] - x).abs().max().item() < 1e-5
    c = ptwt_amd.wavedec3(x, wavelet, level=1)
    cv = [c[0].transpose(1, 2).contiguous().transpose(1, 2)] + [{k: v.clone() for k, v in c[1].items()}]
    rec = ptwt_amd.waverec3(cv, wavelet)
    assert (rec[..., :23, :30, :45] - x).abs().max().item() < 1e-5


# ------------------------------------------------------------------ edge cases and the other BASELINE configs at full size
def test_empty_batch_tiny_and_level0():
    """Empty batches, one-sample signals and level 0 go through without touching a kernel's bounds."""
    for fn, rec, shape in [("wavedec", "waverec", (0, 64)), ("wavedec2", "waverec2", (0, 32, 32)), ("wavedec3", "waverec3", (0, 16, 16, 16))]:
        x = torch.zeros(*shape, device=dev())
        c = getattr(ptwt_amd, fn)(x, "db2", mode="symmetric", level=2)
        assert c[0].shape[0] == 0
        y = getattr(ptwt_amd, rec)(c, "db2")
        assert y.shape[0] == 0
    x = torch.randn(3, 1, device=dev(), dtype=torch.float64)  # N = 1
    check_tree(ptwt_amd.wavedec(x, "db4", mode="zero", level=1), O.wavedec(x.cpu().numpy(), "db4", mode="zero", level=1), TOL64, "N=1")
    x = torch.randn(3, 3, device=dev(), dtype=torch.float64)  # N = 3 < pad: the symmetric extension folds repeatedly
    check_tree(ptwt_amd.wavedec(x, "db4", mode="symmetric", level=1), O.wavedec(x.cpu().numpy(), "db4", mode="symmetric", level=1),
               TOL64, "N=3")
    x = torch.randn(2, 5, 7, device=dev())
    c = ptwt_amd.wavedec2(x, "haar", level=0)
    assert len(c) == 1 and torch.equal(c[0], x)
    x1 = torch.randn(1, 2, 3, device=dev())  # planes smaller than the filter
    check_tree(ptwt_amd.wavedec2(x1, "db3", mode="symmetric", level=1), O.wavedec2(x1.cpu().double().numpy(), "db3", mode="symmetric", level=1), TOL32, "2x3 db3")


def test_full_size_config3_properties():
    """BASELINE configs[2] at full size (8 x 256^3 fp32, db2, level 3, mode zero): oracle check on a sub-volume's worth of
    coefficients (first batch element vs the fp64 oracle), linearity, round trip."""
    torch.manual_seed(3)
    x = torch.randn(8, 256, 256, 256, device=dev())
    c = ptwt_amd.wavedec3(x, "db2", level=3)
    assert tuple(c[0].shape[-3:]) == (34, 34, 34) and tuple(c[-1]["ddd"].shape[-3:]) == (129, 129, 129)
    want = O.wavedec3(to_np(x[0]).astype(np.float64), "db2", level=3)
    got = tuple([c[0][0]] + [{k: v[0] for k, v in d.items()} for d in c[1:]])
    check_tree(got, want, TOL32, "config 3, batch element 0")
    # every other batch element: its first sub-volume (the transform of x[b, :96, :96, :96] agrees with the full one wherever the
    # filters do not reach past the crop: 40^3 / 16^3 / 6^3 coefficients of levels 1 / 2 / 3 from the volume's origin)
    for b in range(1, 8):
        wsub = O.wavedec3(to_np(x[b, :96, :96, :96]).astype(np.float64), "db2", level=3)
        for lvl, n in ((1, 40), (2, 16), (3, 6)):
            for key, ref in wsub[4 - lvl].items():
                g = to_np(c[4 - lvl][key][b, :n, :n, :n])
                assert G.relerr(g, ref[:n, :n, :n]) < TOL32, (b, lvl, key)
        assert G.relerr(to_np(c[0][b, :6, :6, :6]), wsub[0][:6, :6, :6]) < TOL32, b
    y = torch.randn_like(x)
    cz = ptwt_amd.wavedec3(2 * x + y, "db2", level=3)
    cy = ptwt_amd.wavedec3(y, "db2", level=3)
    for (n, a), (_, b), (_, d) in zip(G.flatten_coeffs(cz), G.flatten_coeffs(c), G.flatten_coeffs(cy)):
        assert G.relerr(to_np(a[:2]), to_np((2 * b + d)[:2])) < 2e-6, n
    rec = ptwt_amd.waverec3(c, "db2")
    assert rec.shape == x.shape and (rec - x).abs().max().item() < 1e-5


def test_full_size_config4_slice_properties():
    """BASELINE configs[3], one GPU's 64-image shard (64 x 4096^2 fp32, db8, level 4): oracle check on one image, round trip."""
    torch.manual_seed(4)
    x = torch.randn(64, 4096, 4096, device=dev())
    c = ptwt_amd.wavedec2(x, "db8", level=4)
    assert [tuple(t.shape[-2:]) for t in (c[0], c[1][0], c[2][0], c[3][0], c[4][0])] == [(270, 270), (270, 270), (525, 525), (1035, 1035), (2055, 2055)]
    for b in (0, 31, 63):
        want = O.wavedec2(to_np(x[b]).astype(np.float64), "db8", level=4)
        got = tuple([c[0][b]] + [tuple(t[b] for t in det) for det in c[1:]])
        check_tree(got, want, TOL32, f"config 4, image {b}")
    rec = ptwt_amd.waverec2(c, "db8")
    assert rec.shape == x.shape
    assert G.relerr(to_np(rec[:2]), to_np(x[:2])) < 2e-6 and (rec[60:] - x[60:]).abs().max().item() < 2e-5
    # the reconstruction against the numpy oracle fed the same f32 coefficients (three images, as the analysis above)
    for b in (0, 31, 63):
        cb = tuple([to_np(c[0][b]).astype(np.float64)] + [tuple(to_np(t[b]).astype(np.float64) for t in det) for det in c[1:]])
        assert G.relerr(to_np(rec[b]), O.waverec2(cb, "db8")) < TOL32, f"config 4 reconstruction, image {b}"


def test_full_size_config5_slice_properties():
    """BASELINE configs[4], a 16-image slice (16 x 8192^2 fp16, sym16, level 5, fswavedec2; matrix-core kernel): oracle
    check on the finest level of a crop-independent image (whole image 0 through the fp64 oracle would take minutes:
    the check uses level 1 of image 0 and the deeper levels of a 1024^2 image), sizes, linearity in fp16 tolerance."""
    ptwt_amd.set_half_storage(True)
    try:
        torch.manual_seed(5)
        x = torch.randn(16, 8192, 8192, device=dev()).half()
        c = ptwt_amd.fswavedec2(x, "sym16", level=5)
        assert [tuple(d["dd"].shape[-2:]) for d in c[1:]] == [(286, 286), (541, 541), (1051, 1051), (2071, 2071), (4111, 4111)]
        assert c[0].dtype == torch.float16 and _engine.kernel_id(2, torch.float16, "reflect", 32, 16, (8192, 8192)) == 11
        # level 1 of a 2048-row band of image 0 against the oracle (rows are independent of the rest only through the
        # vertical filter: compare the interior rows of the band)
        band = x[0, 1024:3072].double().cpu().numpy()
        want = O.fswavedec2(band, "sym16", level=1)
        got_dd = c[-1]["dd"][0].double().cpu().numpy()
        # rows 1024..3071 of the image <-> coefficient rows (1024 + 30)/2 .. : interior rows of the band's transform
        lo = 60
        w = want[1]["dd"][lo:-lo]
        g = got_dd[512 + lo: 512 + lo + w.shape[0]]
        assert G.relerr(g, w) < 5e-4
        # levels 1 AND 2 of the same 2048-row band against the numpy oracle (not the on-device f64 transform): level 2 needs the
        # level-1 approximation the kernels actually stored, i.e. rounded to f16 — feed the oracle that
        lvl1 = ptwt_amd.fswavedec2(x[0:1, 1024:3072].contiguous(), "sym16", level=1)
        a1_band = lvl1[0][0].double().cpu().numpy()  # f16-rounded level-1 approximation of the band
        want2 = O.fswavedec2(a1_band, "sym16", level=1)
        got2 = ptwt_amd.fswavedec2(lvl1[0], "sym16", level=1)
        for key in ("ad", "da", "dd"):
            assert G.relerr(got2[1][key][0].double().cpu().numpy(), want2[1][key]) < 5e-4, key
        assert G.relerr(got2[0][0].double().cpu().numpy(), want2[0]) < 5e-4
        want1 = O.fswavedec2(band, "sym16", level=1)
        for key in ("ad", "da", "dd"):
            assert G.relerr(lvl1[1][key][0].double().cpu().numpy(), want1[1][key]) < 5e-4, key
        small = x[1, :1024, :1024].contiguous()
        check = ptwt_amd.fswavedec2(small, "sym16", level=5)
        want = O.fswavedec2(small.double().cpu().numpy(), "sym16", level=5)
        for (n, a), (_, b) in zip(G.flatten_coeffs(check), G.flatten_coeffs(want)):
            assert G.relerr(to_np(a.double()), b) < 2e-3, n  # five levels of f16 storage
    finally:
        ptwt_amd.set_half_storage(False)


def test_full_size_config5_slice_reconstruction():
    """BASELINE configs[4] the other way round, at the benchmark's geometry: `fswaverec2` of a 20-image slice (20 x 8192^2 fp16: image byte
    offsets beyond 2^31; sym16, level 5) with AUTO routing.  (1) the matrix-core synthesis kernel (id 23, csrc/mifwt_dwt2_inv_mfma.hip;
    replaces src/ptwt/separable_conv_transform.py:281-313 / conv_transform_2.py:222-249) serves every level; (2) the finest level of a
    band of coefficient rows against the numpy oracle (hundreds of stacked tiles per panel, units dealt per XCD); (3) round trip
    <= 2e-3 on all images (five levels of f16 storage); (4) images of the batch call bit-identical to one-image calls (first, middle,
    last: the last one lies beyond 2^31 bytes)."""
    ptwt_amd.set_half_storage(True)
    try:
        torch.manual_seed(6)
        nimg = 20
        x = torch.randn(nimg, 8192, 8192, device=dev()).half()
        c = ptwt_amd.fswavedec2(x, "sym16", level=5)
        _engine.level_events = []
        try:
            rec = ptwt_amd.fswaverec2(c, "sym16")
            torch.cuda.synchronize()
            kids = [e[1] for e in _engine.level_events]
        finally:
            _engine.level_events = None
        assert kids == [23] * 5, kids
        assert rec.dtype == torch.float16 and tuple(rec.shape) == tuple(x.shape)
        for b in range(nimg):
            err = float((rec[b].float() - x[b].float()).norm() / x[b].float().norm())
            assert err < 2e-3, (b, err)
        # (4) batch slices against one-image calls
        for b in (0, nimg // 2, nimg - 1):
            cb = tuple([c[0][b : b + 1]] + [{k: v[b : b + 1] for k, v in d.items()} for d in c[1:]])
            one = ptwt_amd.fswaverec2(cb, "sym16")
            assert torch.equal(one[0], rec[b]), b
        # (2) the finest level alone: approximation = the f16 level-1 approximation the kernels stored, details = c[-1]
        lvl1 = ptwt_amd.fswavedec2(x, "sym16", level=1)
        _engine.level_events = []
        try:
            fine = ptwt_amd.fswaverec2(lvl1, "sym16")
            torch.cuda.synchronize()
            kids = [e[1] for e in _engine.level_events]
        finally:
            _engine.level_events = None
        assert kids == [23], kids
        for b, k0 in ((0, 1000), (nimg - 1, 3050)):  # coefficient rows [k0, k0 + 600) of two images, all 4111 columns
            nk, lo = 600, 64
            band = tuple([lvl1[0][b, k0 : k0 + nk].double().cpu().numpy()] + [{k: v[b, k0 : k0 + nk].double().cpu().numpy() for k, v in lvl1[1].items()}])
            want = O.fswaverec2(band, "sym16")  # rows m of it = rows m + 2 k0 of the image's reconstruction, away from the band's ends
            got = fine[b, 2 * k0 + lo : 2 * k0 + want.shape[0] - lo].double().cpu().numpy()
            assert G.relerr(got, want[lo:-lo, :8192]) < 5e-4, (b, k0)
    finally:
        ptwt_amd.set_half_storage(False)


def test_mfma_idwt2_batch_and_single_images_agree():
    """The routing of the f16 long-filter reconstruction looks at ONE image's geometry (a batch took the matrix-core kernel,
    which rounds the intermediate image to f16, where its images one by one took the vector kernel): bit-identical either way."""
    ptwt_amd.set_half_storage(True)
    try:
        rng = np.random.default_rng(77)
        for shape in ((4, 100, 100), (4, 40, 52), (40, 96, 200), (3, 300, 402)):
            x = torch.from_numpy(rng.standard_normal(shape)).half().to(dev())
            c = ptwt_amd.wavedec2(x, "sym16", mode="symmetric", level=1)
            rec = ptwt_amd.waverec2(c, "sym16")
            for b in range(shape[0]):
                one = ptwt_amd.waverec2((c[0][b : b + 1], tuple(t[b : b + 1] for t in c[1])), "sym16")
                assert torch.equal(one[0], rec[b]), (shape, b)
    finally:
        ptwt_amd.set_half_storage(False)


def test_full_size_config5_whole_batch_vs_device_f64_all_levels_and_numpy_oracle_levels_3_to_5():
    """BASELINE configs[4] at its stated size on one GPU: 128 x 8192^2 fp16 (17 GB, 8.6e9 samples: element offsets beyond 2^32),
    sym16, level 5, fswavedec2.  (1) images of the whole-batch call are bit-identical to one-image calls (first, middle, last);
    (2) EVERY sub-band of every level of the last image within 5e-4 (norm-wise) of the fp64 transform of the same fp16-rounded
    approximation — on the device in f64 (streaming axis kernels, pinned against the goldens at 1e-12) for all five levels, and
    against the numpy oracle itself for levels 3-5."""
    ptwt_amd.set_half_storage(True)
    try:
        g = torch.Generator(device=dev()).manual_seed(55)
        x = torch.empty(128, 8192, 8192, device=dev(), dtype=torch.float16)
        for i in range(0, 128, 16):
            x[i : i + 16] = torch.randn(16, 8192, 8192, device=dev(), generator=g).half()
        c = ptwt_amd.fswavedec2(x, "sym16", level=5)
        assert tuple(c[0].shape) == (128, 286, 286)
        assert [tuple(d["dd"].shape) for d in c[1:]] == [(128, n, n) for n in (286, 541, 1051, 2071, 4111)]
        for i in (0, 77, 127):
            ci = ptwt_amd.fswavedec2(x[i : i + 1], "sym16", level=5)
            for (n, a), (_, b) in zip(G.flatten_coeffs(ci), G.flatten_coeffs(c)):
                assert torch.equal(a[0], b[i]), (i, n)
        a = x[127:128]
        for lev in range(1, 6):
            got = ptwt_amd.fswavedec2(a, "sym16", level=1)
            ref = ptwt_amd.fswavedec2(a.double(), "sym16", level=1)
            for (n, u), (_, v) in zip(G.flatten_coeffs(got), G.flatten_coeffs(ref)):
                err = float((u.double() - v).norm() / v.norm())
                assert err < 5e-4, (lev, n, err)
            for k in ("ad", "da", "dd"):  # the level inside the five-level call is this same launch
                assert torch.equal(got[1][k][0], c[6 - lev][k][127]), (lev, k)
            if lev >= 3:
                want = O.fswavedec2(a.double().cpu().numpy(), "sym16", level=1)
                for (n, u), (_, v) in zip(G.flatten_coeffs(got), G.flatten_coeffs(want)):
                    assert G.relerr(to_np(u.double()), v) < 5e-4, (lev, n, "oracle")
            a = got[0]
        assert torch.equal(a[0], c[0][127])
    finally:
        ptwt_amd.set_half_storage(False)


# ---- two analysis levels per launch (mifwt_dwt2_fwd_pair, kernel id 12) ----------------------------------------------
@pytest.fixture(autouse=True)
def _pair_tests_without_pyramid(request):
    """The tests of the two-level kernels switch the three-level launch (kernel id 16, tests/test_gpu_pyramid.py) off: it would
    serve most of their calls first."""
    if request.node.name.startswith("test_pair_kernel"):
        _engine.set_option(_engine.OPT_PYRAMID_MODE, 2)
        yield
        _engine.set_option(_engine.OPT_PYRAMID_MODE, 0)
    else:
        yield


def _pair_vs_single(x, wavelet, mode, level, pair_mode=0):
    """wavedec2 with a pair kernel (pair_mode 0: the library's choice, 1: tiles, 3: rolling strips where they apply)
    against the per-level kernels on the same input: bit-identical."""
    _engine.set_option(_engine.OPT_PAIR_MODE, pair_mode)
    _engine.level_events = []
    try:
        got = ptwt_amd.wavedec2(x, wavelet, mode=mode, level=level)
        kids = [e[1] for e in _engine.level_events]
    finally:
        _engine.level_events = None
        _engine.set_option(_engine.OPT_PAIR_MODE, 0)
    _engine.set_option(_engine.OPT_PAIR_MODE, 2)
    try:
        want = ptwt_amd.wavedec2(x, wavelet, mode=mode, level=level)
    finally:
        _engine.set_option(_engine.OPT_PAIR_MODE, 0)
    gf, wf = G.flatten_coeffs(got), G.flatten_coeffs(want)
    assert [n for n, _ in gf] == [n for n, _ in wf]
    for (n, a), (_, b) in zip(gf, wf):
        assert a.shape == b.shape, (n, a.shape, b.shape)
        assert torch.equal(a, b), f"{wavelet} {mode} L{level} {tuple(x.shape)} {n}: max diff {(a - b).abs().max().item():.3e}"
    return kids


@pytest.mark.parametrize("pair_mode", [0, 1, 3])
@pytest.mark.parametrize("wavelet", ["haar", "db2", "db3", "db4"])
@pytest.mark.parametrize("mode", ["reflect", "zero", "constant", "symmetric"])
def test_pair_kernel_bit_identical_to_per_level(wavelet, mode, pair_mode):
    g = torch.Generator().manual_seed(11)
    for shape, level in [((3, 200, 300), 2), ((2, 257, 131), 2), ((2, 333, 517), 3), ((1, 1024, 1024), 4), ((5, 128, 136), 2),
                         ((2, 61, 140), 2), ((1, 1030, 129), 2)]:
        x = torch.randn(*shape, generator=g, dtype=torch.float32).to(dev())
        kids = _pair_vs_single(x, wavelet, mode, level, pair_mode)
        assert kids[0] == _engine.KID_PAIR, (shape, kids)


@pytest.mark.parametrize("pair_mode", [1, 3])
@pytest.mark.parametrize("rows", [4, 6, 8, 12, 16, 40, 64])
def test_pair_kernel_tile_heights(rows, pair_mode):
    """OPT_PAIR_ROWS: level-2 rows per tile (tile kernel) / per strip segment (rolling kernel, rounded up to 8)."""
    g = torch.Generator().manual_seed(12)
    x = torch.randn(2, 389, 611, generator=g, dtype=torch.float32).to(dev())
    _engine.set_option(_engine.OPT_PAIR_ROWS, rows)
    try:
        for wavelet in ("db1", "db4"):
            for mode in ("reflect", "symmetric", "zero"):
                kids = _pair_vs_single(x, wavelet, mode, 2, pair_mode)
                assert kids == [_engine.KID_PAIR]
    finally:
        _engine.set_option(_engine.OPT_PAIR_ROWS, 0)


def test_pair_kernel_vs_oracle_and_fallbacks():
    g = torch.Generator().manual_seed(13)
    x = torch.randn(2, 300, 260, generator=g, dtype=torch.float32)
    got = ptwt_amd.wavedec2(x.to(dev()), "db4", mode="symmetric", level=3)
    want = O.wavedec2(x.numpy().astype(np.float64), "db4", mode="symmetric", level=3)
    check_tree(got, want, TOL32, "pair db4 symmetric")
    # periodic extension, long filters, narrow planes and f64 are served level by level
    for xx, wavelet, mode in [(x, "db4", "periodic"), (x, "db8", "reflect"), (x[..., :100], "db2", "reflect"), (x.double(), "db2", "zero")]:
        _engine.level_events = []
        try:
            ptwt_amd.wavedec2(xx.to(dev()), wavelet, mode=mode, level=2)
            kids = [e[1] for e in _engine.level_events]
        finally:
            _engine.level_events = None
        assert _engine.KID_PAIR not in kids and len(kids) == 2, (wavelet, mode, kids)
    # non-contiguous rows (a column slice of a wider image) and a strided batch
    wide = torch.randn(4, 300, 700, generator=g, dtype=torch.float32).to(dev())
    view = wide[::2, :, 100:500]
    kids = _pair_vs_single(view, "db3", "reflect", 2)
    assert kids == [_engine.KID_PAIR]
    check_tree(ptwt_amd.wavedec2(view, "db3", level=2), O.wavedec2(view.cpu().numpy().astype(np.float64), "db3", mode="reflect", level=2), TOL32)


@pytest.mark.parametrize("mode", ["reflect", "zero", "constant", "symmetric"])
@pytest.mark.parametrize("wavelet", ["haar", "db2", "db3", "db4"])
def test_pair_kernels_vs_oracle_at_odd_extents(wavelet, mode):
    """The two-levels-per-launch kernels against the fp64 ORACLE (not only against the per-level kernels): analysis pair (tiles and
    rolling strips) and synthesis pair, odd extents at both levels."""
    g = torch.Generator().manual_seed(14)
    x = torch.randn(2, 211, 307, generator=g, dtype=torch.float32)
    want = O.wavedec2(x.numpy().astype(np.float64), wavelet, mode=mode, level=2)
    for pair_mode in (1, 3):
        _engine.set_option(_engine.OPT_PAIR_MODE, pair_mode)
        _engine.level_events = []
        try:
            got = ptwt_amd.wavedec2(x.to(dev()), wavelet, mode=mode, level=2)
            kids = [e[1] for e in _engine.level_events]
        finally:
            _engine.level_events = None
            _engine.set_option(_engine.OPT_PAIR_MODE, 0)
        assert kids == [_engine.KID_PAIR], kids
        check_tree(got, want, TOL32, f"pair {wavelet} {mode} mode {pair_mode}")
    coeffs = [torch.from_numpy(want[0]).float().to(dev())] + [tuple(torch.from_numpy(np.asarray(t)).float().to(dev()) for t in d) for d in want[1:]]
    _engine.level_events = []
    try:
        rec = ptwt_amd.waverec2(coeffs, wavelet)
        kids = [e[1] for e in _engine.level_events]
    finally:
        _engine.level_events = None
    assert kids == [_engine.KID_INV_PAIR], kids
    ref = O.waverec2(want, wavelet)
    assert rec.shape == ref.shape and G.relerr(to_np(rec), ref) < 2e-6


def test_pair_kernel_full_size_config2_round_trip():
    x = torch.randn(64, 1024, 1024, dtype=torch.float32, device=dev())
    _engine.level_events = []
    try:
        c = ptwt_amd.wavedec2(x, "db4", level=3)
        kids = [e[1] for e in _engine.level_events]
    finally:
        _engine.level_events = None
    assert kids == [_engine.KID_PAIR, 7], kids
    rec = ptwt_amd.waverec2(c, "db4")
    assert (rec - x).abs().max().item() < 5e-6


# ---- two synthesis levels per launch (mifwt_dwt2_inv_pair, kernel id 13) ---------------------------------------------
@pytest.mark.parametrize("wavelet", ["haar", "db2", "db3", "db4"])
def test_idwt_pair_kernel_bit_identical_to_per_level(wavelet):
    """waverec2 with the two-level synthesis kernel against the per-level kernels on the same coefficients (odd and even
    extents, i.e. with and without the reference's end-crop between levels; strided batch): bit-identical, and the round
    trip closes."""
    g = torch.Generator().manual_seed(21)
    for shape, level in [((3, 200, 300), 2), ((2, 257, 131), 2), ((2, 333, 517), 3), ((1, 1024, 1024), 4), ((5, 128, 136), 2), ((2, 1030, 129), 3)]:
        x = torch.randn(*shape, generator=g, dtype=torch.float32).to(dev())
        for mode in ("reflect", "zero"):
            c = ptwt_amd.wavedec2(x, wavelet, mode=mode, level=level)
            _engine.level_events = []
            _engine.set_option(_engine.OPT_PYRAMID_MODE, 2)  # (planes from 512 columns on would go to the streaming launch, kernel id 22)
            try:
                got = ptwt_amd.waverec2(c, wavelet)
                kids = [e[1] for e in _engine.level_events]
            finally:
                _engine.level_events = None
                _engine.set_option(_engine.OPT_PYRAMID_MODE, 0)
            _engine.set_option(_engine.OPT_PAIR_MODE, 2)
            try:
                want = ptwt_amd.waverec2(c, wavelet)
            finally:
                _engine.set_option(_engine.OPT_PAIR_MODE, 0)
            assert kids[-1] == _engine.KID_INV_PAIR, (shape, wavelet, kids)  # the finest two levels go as a pair
            assert got.shape == want.shape and torch.equal(got, want), (shape, wavelet, mode, (got - want).abs().max().item())
            assert (got[..., : shape[-2], : shape[-1]] - x).abs().max().item() < 5e-6


def test_idwt_pair_fallbacks_and_views():
    g = torch.Generator().manual_seed(22)
    x = torch.randn(2, 300, 260, generator=g, dtype=torch.float32)
    # f64 and long filters are served level by level, small planes by the whole-reconstruction launch
    for xx, wavelet in [(x.double(), "db2"), (x, "db8"), (x[..., :60, :60], "db2")]:
        c = ptwt_amd.wavedec2(xx.to(dev()), wavelet, level=2)
        _engine.level_events = []
        try:
            ptwt_amd.waverec2(c, wavelet)
            kids = [e[1] for e in _engine.level_events]
        finally:
            _engine.level_events = None
        assert _engine.KID_INV_PAIR not in kids and (len(kids) == 2 or kids == [_engine.KID_INV_SMALL]), (wavelet, kids)
        assert (kids == [_engine.KID_INV_SMALL]) == (xx.shape[-1] == 60), (wavelet, kids)
    # coefficients that are views with foreign strides (channels-last style batch) still reconstruct exactly as per level
    c = ptwt_amd.wavedec2(x.to(dev()), "db3", level=2)
    cv = [c[0].transpose(0, 1).contiguous().transpose(0, 1)] + [type(d)(*(t.clone() for t in d)) for d in c[1:]]
    got = ptwt_amd.waverec2(cv, "db3")
    _engine.set_option(_engine.OPT_PAIR_MODE, 2)
    try:
        want = ptwt_amd.waverec2(cv, "db3")
    finally:
        _engine.set_option(_engine.OPT_PAIR_MODE, 0)
    assert torch.equal(got, want)
    check_tree([got], [O.waverec2(O.wavedec2(x.numpy().astype(np.float64), "db3", level=2), "db3")], 2e-6, "idwt pair vs oracle")


def test_idwt_pair_serves_separable_containers():
    """fswaverec2 crops the running approximation to the next level's detail shape (separable_conv_transform.py:94-97);
    in the two-level kernel that crop is the extent of the approximation tile."""
    g = torch.Generator().manual_seed(23)
    for shape in [(2, 257, 131), (3, 200, 301), (1, 515, 515)]:
        x = torch.randn(*shape, generator=g, dtype=torch.float32).to(dev())
        for wavelet in ("db2", "db4"):
            c = ptwt_amd.fswavedec2(x, wavelet, level=3)
            _engine.level_events = []
            _engine.set_option(_engine.OPT_PYRAMID_MODE, 2)  # (planes from 512 columns on would go to the streaming launch, kernel id 22)
            try:
                got = ptwt_amd.fswaverec2(c, wavelet)
                kids = [e[1] for e in _engine.level_events]
            finally:
                _engine.level_events = None
                _engine.set_option(_engine.OPT_PYRAMID_MODE, 0)
            _engine.set_option(_engine.OPT_PAIR_MODE, 2)
            try:
                want = ptwt_amd.fswaverec2(c, wavelet)
            finally:
                _engine.set_option(_engine.OPT_PAIR_MODE, 0)
            assert kids[-1] == _engine.KID_INV_PAIR, (shape, wavelet, kids)
            assert torch.equal(got, want)
            assert (got[..., : shape[-2], : shape[-1]] - x).abs().max().item() < 5e-6


# ---- the deep levels of a 1-D decomposition in one launch (mifwt_dwt1_fwd_tail, kernel id 14) --------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_dwt1_tail_fusion_vs_oracle_and_per_level(dtype):
    """wavedec with the fused deep levels against the fp64 oracle and against the per-level kernels: every boundary mode,
    filters up to 32 taps, rows longer than the kernel's LDS limit (the first levels then run one by one), levels shorter
    than the filter (the index map folds repeatedly), batch views with a row stride."""
    tol = TOL64 if dtype == torch.float64 else TOL32
    rng = np.random.default_rng(31)
    cases = [("haar", (1, 4096), None), ("db4", (3, 5000), None), ("db5", (2, 40000), 8), ("sym16", (2, 3000), None), ("db2", (5, 33), None),
             ("db8", (4, 1000), 6)]
    for wavelet, shape, level in cases:
        x = rng.standard_normal(shape)
        xg = torch.from_numpy(x).to(dtype).to(dev())
        for mode in MODES:
            try:
                want = O.wavedec(x, wavelet, mode=mode, level=level)
            except RuntimeError:
                continue
            _engine.level_events = []
            try:
                got = ptwt_amd.wavedec(xg, wavelet, mode=mode, level=level)
                kids = [e[1] for e in _engine.level_events]
            finally:
                _engine.level_events = None
            if len(want) > 2:
                assert kids[-1] == _engine.KID_TAIL, (wavelet, shape, mode, kids)
            check_tree(got, want, tol, f"tail {wavelet} {shape} {mode}")
            _engine.set_option(_engine.OPT_PAIR_MODE, 2)
            try:
                single = ptwt_amd.wavedec(xg, wavelet, mode=mode, level=level)
            finally:
                _engine.set_option(_engine.OPT_PAIR_MODE, 0)
            for a, b in zip(got, single):
                assert a.shape == b.shape and G.relerr(to_np(a), to_np(b)) < (1e-13 if dtype == torch.float64 else 2e-6)
    # rows of a wider tensor (row stride != length) and the round trip
    wide = torch.randn(6, 9000, device=dev(), dtype=dtype)
    view = wide[::2, 100:8100]
    c = ptwt_amd.wavedec(view, "db3", level=9)
    check_tree(c, O.wavedec(view.cpu().double().numpy(), "db3", level=9), tol, "tail strided rows")
    rec = ptwt_amd.waverec(c, "db3")
    assert (rec[..., :8000] - view).abs().max().item() < (1e-12 if dtype == torch.float64 else 5e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_idwt1_tail_fusion_vs_oracle_and_per_level(dtype):
    """waverec with the fused coarse levels (mifwt_dwt1_inv_tail, kernel id 15) against the fp64 oracle and the per-level
    kernels: odd and even lengths (with and without the reference's end-crop between levels), rows that outgrow the LDS
    limit mid-way (the remaining levels run one by one), levels shorter than the filter."""
    tol = 1e-12 if dtype == torch.float64 else 2e-6
    rng = np.random.default_rng(41)
    for wavelet, shape, level in [("haar", (1, 4096), None), ("db4", (3, 5001), None), ("db5", (2, 40003), 8), ("sym16", (2, 3000), None),
                                  ("db2", (5, 33), None), ("db8", (4, 1000), 6)]:
        x = rng.standard_normal(shape)
        want_c = O.wavedec(x, wavelet, level=level)
        want = O.waverec(want_c, wavelet)
        cg = [torch.from_numpy(c).to(dtype).to(dev()) for c in want_c]
        _engine.level_events = []
        try:
            got = ptwt_amd.waverec(cg, wavelet)
            kids = [e[1] for e in _engine.level_events]
        finally:
            _engine.level_events = None
        if dtype == torch.float64 or shape[-1] < 4096:
            assert kids[0] == _engine.KID_INV_TAIL, (wavelet, shape, kids)
        else:  # f32: few long rows are cut into chunks instead (mifwt_dwt1_inv_long, tests/test_gpu_long1d.py)
            assert _engine.KID_INV_LONG in kids or _engine.KID_INV_TAIL in kids, (wavelet, shape, kids)
        assert got.shape == want.shape and G.relerr(to_np(got), want) < tol, (wavelet, shape)
        _engine.set_option(_engine.OPT_PAIR_MODE, 2)
        try:
            single = ptwt_amd.waverec(cg, wavelet)
        finally:
            _engine.set_option(_engine.OPT_PAIR_MODE, 0)
        assert G.relerr(to_np(got), to_np(single)) < (1e-13 if dtype == torch.float64 else 2e-6)


def test_f16_storage_outside_the_fused_envelopes():
    """f16 storage on geometries no fused kernel takes (planes shorter than the filter, filter lengths the streaming kernels are
    not instantiated for, 3-D) and its backward with a non-zero boundary mode: the generic per-axis passes serve them (f16
    storage, f32 arithmetic) instead of MIFWT_ERR_UNSUPPORTED.  Oracle: fp64 transform of the f16-quantised input, 2e-3 per level
    (the passes of an N-D level round to f16 in between, like every f16 path)."""
    ptwt_amd.set_half_storage(True)
    try:
        torch.manual_seed(21)
        for fn, ofn, shape, wavelet, mode, level in [("wavedec2", O.wavedec2, (2, 5, 40), "db4", "symmetric", 1),
                                                     ("wavedec2", O.wavedec2, (2, 70, 66), "db11", "reflect", 2),
                                                     ("wavedec3", O.wavedec3, (1, 20, 24, 18), "db2", "constant", 2),
                                                     ("wavedec", O.wavedec, (3, 300), "coif5", "periodic", 2)]:
            x = torch.randn(*shape, device=dev()).half()
            got = getattr(ptwt_amd, fn)(x, wavelet, mode=mode, level=level)
            want = ofn(x.double().cpu().numpy(), wavelet, mode=mode, level=level)
            for (n, a), (_, b) in zip(G.flatten_coeffs(got), G.flatten_coeffs(want)):
                assert a.dtype == torch.float16 and G.relerr(to_np(a.double()), b) < 2e-3 * level, (fn, wavelet, n)
        # backward through a reflect-mode level in f16 storage: adjoint identity <A x, w> = <x, A^T w> in f16 tolerance
        x = torch.randn(2, 48, 52, device=dev()).half().requires_grad_(True)
        c = ptwt_amd.wavedec2(x, "db3", mode="reflect", level=1)
        ws = [torch.randn_like(t) for _, t in G.flatten_coeffs(c)]
        lhs = sum((t.float() * w.float()).sum() for (_, t), w in zip(G.flatten_coeffs(c), ws))
        (gx,) = torch.autograd.grad(sum((t * w).sum() for (_, t), w in zip(G.flatten_coeffs(c), ws)), x)
        rhs = (x.detach().float() * gx.float()).sum()
        assert abs(float(lhs.detach() - rhs)) < 5e-3 * float(lhs.detach().abs() + rhs.abs() + 1.0)
    finally:
        ptwt_amd.set_half_storage(False)
