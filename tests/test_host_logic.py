"""CPU tests of the host layer (no GPU): argument handling, containers, error behaviour, C-ABI surface.

The level engine is replaced by the oracle (tests/_oracle_engine.py) so the whole Python glue of
``ptwt_amd`` runs against the reference's golden outputs here; the HIP path itself is tested in
tests/test_gpu_parity.py (``-m gpu``).
"""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import ptwt_amd
from ptwt_amd import _engine, _fwt, _wavelets
from tests import _golden as G
from tests._oracle_engine import OracleLevelEngine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def oracle_engine(monkeypatch):
    from ptwt_amd import stationary_transform as st
    from tests import _oracle_engine as oe

    monkeypatch.setattr(_engine, "ENGINE", OracleLevelEngine())
    monkeypatch.setattr(st, "_level_fwd", oe.swt_level_fwd)
    monkeypatch.setattr(st, "_level_inv", oe.swt_level_inv)


def test_public_surface_matches_reference():
    """The ten functions + type aliases of reference src/ptwt/__init__.py:12-19 (conv-FWT path)."""
    for name in ["wavedec", "waverec", "wavedec2", "waverec2", "wavedec3", "waverec3", "fswavedec2", "fswavedec3",
                 "fswaverec2", "fswaverec3", "Wavelet", "WaveletDetailTuple2d", "WaveletCoeff2d", "WaveletCoeffNd",
                 "WaveletCoeff2dSeparable", "WaveletDetailDict", "WaveletTensorTuple"]:
        assert hasattr(ptwt_amd, name), name
    import inspect

    sig = inspect.signature(ptwt_amd.wavedec2)
    assert [p.name for p in sig.parameters.values()] == ["data", "wavelet", "mode", "level", "axes"]
    assert sig.parameters["mode"].default == "reflect" and sig.parameters["axes"].default == (-2, -1)
    assert sig.parameters["mode"].kind is inspect.Parameter.KEYWORD_ONLY
    assert inspect.signature(ptwt_amd.wavedec3).parameters["mode"].default == "zero"
    assert inspect.signature(ptwt_amd.wavedec).parameters["axis"].default == -1
    assert inspect.signature(ptwt_amd.fswavedec2).parameters["axes"].default is None
    from ptwt_amd.conv_transform_2 import wavedec2  # same module names as the reference
    from ptwt_amd.separable_conv_transform import fswaverec3  # noqa: F401

    assert wavedec2 is ptwt_amd.wavedec2


def test_no_cpu_fallback():
    """A CPU tensor must fail loudly: the product path has no CPU/eager fallback."""
    with pytest.raises(RuntimeError, match="ROCm device"):
        ptwt_amd.wavedec2(torch.zeros(2, 16, 16), "haar", level=1)
    with pytest.raises(RuntimeError, match="ROCm device"):
        ptwt_amd.waverec([torch.zeros(1, 8), torch.zeros(1, 8)], "haar")


def test_builtin_wavelets_equal_pywt_table():
    import json

    with open(os.path.join(G.GOLDEN, "pywt_filter_banks.json")) as f:
        gold = json.load(f)
    names = _wavelets.wavelist()
    assert len(names) == 106
    for n in names:
        w = _wavelets.as_wavelet(n)
        assert list(w.dec_lo) == gold[n]["dec_lo"] and list(w.rec_hi) == gold[n]["rec_hi"]
        assert len(w) == len(gold[n]["dec_lo"])
    assert _wavelets.host_taps("db1") == _wavelets.host_taps("haar")
    with pytest.raises(ValueError):
        _wavelets.as_wavelet("not-a-wavelet")
    assert _wavelets.dwt_max_level(4096, 2) == 12 and _wavelets.dwt_max_level(5, 8) == 0
    assert _wavelets.dwtn_max_level([1024, 1024], 8) == 7


def test_wavelet_argument_forms(oracle_engine):
    """str | object with .filter_bank | 4-tuple of tensors (reference src/ptwt/_util.py:71-126)."""
    x = torch.randn(2, 40, dtype=torch.float64)
    w = _wavelets.as_wavelet("db3")
    a = ptwt_amd.wavedec(x, "db3", level=2)
    b = ptwt_amd.wavedec(x, w, level=2)
    c = ptwt_amd.wavedec(x, ptwt_amd.WaveletTensorTuple.from_wavelet(w, torch.float64), level=2)

    class Bank:
        filter_bank = w.filter_bank

        def __len__(self):
            return 6

    d = ptwt_amd.wavedec(x, Bank(), level=2)
    for u, v, y, z in zip(a, b, c, d):
        assert torch.equal(u, v) and torch.equal(u, y) and torch.equal(u, z)


def test_kat_ripples_haar(oracle_engine):
    """Reference tests/test_convolution_fwt.py:98-118 through the ptwt_amd host layer."""

    class Haar:
        filter_bank = ([0.5, 0.5], [-0.5, 0.5], [0.5, 0.5], [0.5, -0.5])

        def __len__(self):
            return 2

    c = ptwt_amd.wavedec(torch.tensor([56.0, 40.0, 8.0, 24.0, 48.0, 48.0, 40.0, 16.0]), Haar(), level=3)
    assert c[0].item() == 35.0 and c[1].item() == -3.0
    assert c[2].tolist() == [16.0, 10.0] and c[3].tolist() == [8.0, -8.0, 0.0, 12.0]


@pytest.mark.parametrize("case", G.ref_cases(), ids=lambda c: "%s-%s-%s-%s" % (c["key"], c["fn"], c["wavelet"], c["dtype"]))
def test_host_layer_vs_reference_goldens(case, oracle_engine):
    """Containers, key order, shapes, dtype, axes/fold handling and the synthesis trims, on every reference
    golden case (the arithmetic is the oracle's; the plumbing is the product's)."""
    z, _ = G.load("ptwt_ref.npz")
    k = case["key"]
    x = torch.from_numpy(z[k + "_x"])
    kw = {a: (tuple(v) if isinstance(v, list) else v) for a, v in case["kw"].items()}
    coeffs = getattr(ptwt_amd, case["fn"])(x, case["wavelet"], **kw)
    assert isinstance(coeffs, list if case["fn"] == "wavedec" else tuple)
    flat = G.flatten_coeffs(coeffs)
    assert [n for n, _ in flat] == case["names"]
    tol = 1e-12 if case["dtype"] == "float64" else 1e-6
    for name, val in flat:
        want = z["%s_%s" % (k, name)]
        assert val.dtype == x.dtype and tuple(val.shape) == want.shape
        assert G.relerr(val.numpy(), want) < tol
    if case["fn"] == "wavedec2":
        assert all(isinstance(c, ptwt_amd.WaveletDetailTuple2d) for c in coeffs[1:])
    rkw = {a: v for a, v in kw.items() if a in ("axis", "axes")}
    rec = getattr(ptwt_amd, case["rec"])(coeffs, case["wavelet"], **rkw)
    want = z[k + "_rec"]
    assert tuple(rec.shape) == want.shape
    assert G.relerr(rec.numpy(), want) < (1e-11 if case["dtype"] == "float64" else 2e-6)


def test_error_behaviour(oracle_engine):
    """Error types pinned by the reference (reference tests/test_convolution_fwt.py:347-402,
    tests/test_convolution_fwt_3.py:167-178)."""
    x = torch.randn(3, 16, 16, dtype=torch.float64)
    with pytest.raises(ValueError):
        ptwt_amd.wavedec2(x.to(torch.float16), "haar")  # unsupported dtype
    with pytest.raises(ValueError):
        ptwt_amd.wavedec2(x.to(torch.int32), "haar")
    with pytest.raises(ValueError):
        ptwt_amd.wavedec2(torch.randn(16, dtype=torch.float64), "haar")  # too few dims
    with pytest.raises(ValueError):
        ptwt_amd.wavedec3(torch.randn(4, 4, dtype=torch.float64), "haar")
    with pytest.raises(ValueError):
        ptwt_amd.wavedec2(x, "haar", mode="bogus", level=1)
    with pytest.raises(ValueError):
        ptwt_amd.wavedec2(x, "haar", axes=(1, 1))  # repeated axis
    with pytest.raises(ValueError):
        ptwt_amd.wavedec2(x, "haar", axes=(0, 1, 2))  # wrong count
    with pytest.raises(ValueError):
        ptwt_amd.wavedec2(x, "haar", axes=1)  # int for a 2-D transform
    with pytest.raises(RuntimeError):
        ptwt_amd.wavedec(torch.randn(1, 6, dtype=torch.float64), "db4", mode="reflect", level=1)  # pad >= N
    ptwt_amd.wavedec(torch.randn(1, 4, dtype=torch.float64), "db4", mode="symmetric", level=1)  # tolerated
    c = ptwt_amd.wavedec2(x, "db2", level=2)
    with pytest.raises(ValueError):
        ptwt_amd.waverec2((c[0], tuple(c[1][:2])), "db2")  # not a 3-tuple
    with pytest.raises(ValueError):
        ptwt_amd.waverec2((c[0], c[2]), "db2")  # shape mismatch inside a level
    with pytest.raises(ValueError):
        ptwt_amd.waverec2((c[0].to(torch.float32), c[1], c[2]), "db2")  # dtype mismatch
    with pytest.raises(ValueError):
        ptwt_amd.waverec2(([1, 2], c[1]), "db2")  # first element not a tensor
    with pytest.raises(AssertionError):
        ptwt_amd.waverec2((c[0], c[1], c[2]), "db4")  # wavelet does not match the coefficients
    c3 = ptwt_amd.wavedec3(torch.randn(8, 8, 8, dtype=torch.float64), "haar", level=1)
    bad = dict(c3[1])
    bad.pop("ddd")
    with pytest.raises(ValueError):
        ptwt_amd.waverec3((c3[0], bad), "haar")
    with pytest.raises(ValueError):
        ptwt_amd.fswaverec2((c3[0], [1, 2, 3]), "haar")
    # level = 0 returns the input as the only coefficient, list for 1-D / tuple otherwise
    assert isinstance(ptwt_amd.wavedec(x, "haar", level=0), list)
    out = ptwt_amd.wavedec2(x, "haar", level=0)
    assert isinstance(out, tuple) and len(out) == 1 and torch.equal(out[0], x)


def test_fswaverec_does_not_mutate_input(oracle_engine):
    x = torch.randn(2, 17, 18, dtype=torch.float64)
    c = ptwt_amd.fswavedec2(x, "db2", level=2)
    keys = [list(d.keys()) for d in c[1:]]
    rec = ptwt_amd.fswaverec2(c, "db2")
    assert [list(d.keys()) for d in c[1:]] == keys == [["da", "ad", "dd"]] * 2
    assert torch.allclose(rec[..., :17, :18], x, atol=1e-10)


# ------------------------------------------------------------------------------------------ C ABI (no GPU)
def _header_symbols():
    text = open(os.path.join(ROOT, "include", "mifwt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mifwt_[a-z_]+)\s*\(", text)))


def test_cabi_library_exports_every_declared_symbol():
    lib = _engine.load_library()
    syms = _header_symbols()
    assert {"mifwt_dwt_fwd", "mifwt_dwt_inv", "mifwt_workspace_bytes", "mifwt_kernel_id", "mifwt_strerror",
            "mifwt_abi_version", "mifwt_set_option"} <= set(syms)
    for s in syms:
        assert getattr(lib, s) is not None
    assert lib.mifwt_abi_version() == _engine.ABI_VERSION == 3 and lib.mifwt_launch_count(99) == 0
    assert lib.mifwt_strerror(0) == b"ok" and b"argument" in lib.mifwt_strerror(-1)


def test_product_library_has_no_result_breaking_switches():
    """The measurement switches of MIFWT_OPT_DEBUG that break results (no stores / no loads / no deep levels ...), the experiment word and the
    profiling instances are compiled into -DMIFWT_DIAG builds only: the product library refuses them (csrc/mifwt_common.h), keeps the
    routing bits (alternative code paths with the same results), and no longer exports round 3's handover entry points."""
    lib = _engine.load_library()
    if os.environ.get("MIFWT_LIB"):
        pytest.skip("an experiment build may be a diagnostics build")
    for bits in (1, 2, 4, 16, 32, 128, 256, 2048, 65536, 1 | 1024):
        assert lib.mifwt_set_option(_engine.OPT_DEBUG, bits) == -2  # MIFWT_ERR_UNSUPPORTED
    for bits in (8, 64, 512, 1024, 4096, 8192, 1 << 19, 1 << 20, 1024 | 4096):
        assert lib.mifwt_set_option(_engine.OPT_DEBUG, bits) == 0
    assert lib.mifwt_set_option(_engine.OPT_DEBUG, 0) == 0
    assert lib.mifwt_set_option(_engine.OPT_EXP, 1) == -2 and lib.mifwt_set_option(_engine.OPT_EXP, 0) == 0
    lib.mifwt_pyr_profile_buffer.argtypes = [ctypes.c_void_p]
    assert lib.mifwt_pyr_profile_buffer(ctypes.c_void_p(4096)) == -2 and lib.mifwt_pyr_profile_buffer(None) == 0
    for gone in ("mifwt_dwt2_fwd_pyramid_ws", "mifwt_dwt2_fwd_pyramid_workspace"):
        assert not hasattr(lib, gone)


def test_cabi_descriptor_validation_and_dispatch():
    """Argument validation and kernel selection run on the host, so they are testable without a GPU."""
    lib = _engine.load_library()
    d = _engine.LevelDesc()
    assert lib.mifwt_kernel_id(ctypes.byref(d), 0) == -1  # ndim = 0
    assert _engine.kernel_id(2, torch.float32, "reflect", 8, 64, (1024, 1024)) == 16  # one level through the streaming multi-level kernel (round 4)
    assert _engine.kernel_id(2, torch.float32, "reflect", 8, 64, (640, 640)) == 7  # fused 2-D analysis, LDS tiles
    assert _engine.kernel_id(2, torch.float32, "reflect", 16, 64, (4096, 4096)) == 1  # fused 2-D analysis, streaming
    assert _engine.kernel_id(2, torch.float32, "reflect", 16, 64, (1035, 1035)) == 7
    assert _engine.kernel_id(2, torch.float32, "reflect", 8, 64, (1024, 1024), direction=1) == 22  # one level through the streaming multi-level kernel (round 4)
    assert _engine.kernel_id(2, torch.float32, "reflect", 8, 64, (2048, 2048), direction=1) == 2  # fused 2-D synthesis, streaming wave strips
    assert _engine.kernel_id(2, torch.float64, "reflect", 8, 64, (1024, 1024)) == 7  # f64, L <= 16: LDS tiles in double
    assert _engine.kernel_id(2, torch.float64, "reflect", 8, 64, (1024, 1024), direction=1) == 8
    assert _engine.kernel_id(2, torch.float64, "reflect", 20, 4, (512, 512)) == 3    # f64, long filter: streaming axis passes
    assert _engine.kernel_id(2, torch.float32, "reflect", 32, 4, (512, 512)) == 7    # L = 32 analysis -> LDS-tile kernel
    assert _engine.kernel_id(2, torch.float16, "reflect", 32, 4, (512, 512)) == 11   # f16 + long filter: matrix cores
    assert _engine.kernel_id(2, torch.float16, "reflect", 8, 4, (512, 512)) == 7     # f16, short filter: LDS tiles
    assert _engine.kernel_id(2, torch.float32, "reflect", 32, 4, (512, 512), direction=1) == 8  # synthesis: LDS tiles
    assert _engine.kernel_id(2, torch.float32, "reflect", 8, 64, (515, 515), direction=1) == 8   # small plane: tiles
    assert _engine.kernel_id(1, torch.float64, "zero", 2, 1, (4096,)) == 3
    assert _engine.kernel_id(1, torch.float32, "zero", 8, 1, (4096,), direction=1) == 4
    assert _engine.kernel_id(3, torch.float32, "zero", 4, 8, (256, 256, 256)) == 24  # depth-walking fused kernel (big volumes)
    assert _engine.kernel_id(3, torch.float32, "zero", 4, 4, (128, 128, 128)) == 9   # fully fused LDS-brick kernel
    assert _engine.kernel_id(3, torch.float32, "zero", 4, 8, (128, 128, 128)) == 24  # (round 6: eight volumes of 2^21 samples on walk too)
    assert _engine.kernel_id(3, torch.float32, "zero", 12, 8, (256, 256, 256)) == 5  # fused planes + depth pass
    assert _engine.kernel_id(3, torch.float32, "zero", 4, 8, (256, 256, 256), direction=1) == 25  # depth-walking fused synthesis (from ~1 M outputs)
    assert _engine.kernel_id(3, torch.float32, "zero", 4, 8, (64, 64, 64), direction=1) == 10  # fully fused LDS-brick synthesis
    assert _engine.kernel_id(3, torch.float32, "zero", 8, 8, (256, 256, 256), direction=1) == 25 and _engine.kernel_id(3, torch.float32, "zero", 8, 8, (64, 64, 64), direction=1) == 10 and _engine.kernel_id(3, torch.float32, "zero", 10, 8, (256, 256, 256), direction=1) == 6   # ten taps: fused planes + depth pass
    assert _engine.kernel_id(3, torch.float64, "zero", 4, 8, (256, 256, 256), direction=1) != 10
    assert _engine.kernel_id(2, torch.float32, "reflect", 102, 4, (512, 512)) == 0   # coif17 -> generic passes
    assert _engine.kernel_id(2, torch.float32, "reflect", 22, 4, (512, 512)) == 0    # L not in the streaming set
    # a level described without strides (not unit innermost) is generic and needs scratch, the fused 2-D level none
    d = _engine.LevelDesc()
    d.ndim, d.dtype, d.mode, d.filt_len, d.batch = 3, 0, 0, 4, 2
    for a in range(3):
        d.sig_extent[a], d.coef_extent[a] = 16, 9
    assert lib.mifwt_workspace_bytes(ctypes.byref(d), 0) == 4 * 2 * (2 * 16 * 16 * 9 + 4 * 16 * 9 * 9)
    d.coef_extent[1] = 8  # inconsistent with (N + L - 1) // 2
    assert lib.mifwt_kernel_id(ctypes.byref(d), 0) == -1
    assert lib.mifwt_dwt_fwd(ctypes.byref(d), None, None, None, None, None, None, 0, None) == -1


# ---- multi-level entry points: envelopes and argument checks run on the host ----------------------------------------------
def _dense_desc(dtype_id, mode, flen, batch, sig, coef):
    d = _engine.LevelDesc()
    d.ndim, d.dtype, d.mode, d.filt_len, d.batch = 2, dtype_id, _engine.MODE_IDS[mode], flen, batch
    for a in range(2):
        d.sig_extent[a], d.coef_extent[a] = sig[a], coef[a]
    d.sig_stride[0], d.sig_stride[1], d.sig_stride[2] = sig[0] * sig[1], sig[1], 1
    for st in (d.approx_stride, d.detail_stride):
        st[0], st[1], st[2] = 4 * coef[0] * coef[1], coef[1], 1
    return d


def _analysis_pair_descs(dtype_id, mode, flen, batch, shape):
    c1 = [(n + flen - 1) // 2 for n in shape]
    c2 = [(n + flen - 1) // 2 for n in c1]
    return _dense_desc(dtype_id, mode, flen, batch, shape, c1), _dense_desc(dtype_id, mode, flen, batch, c1, c2)


def test_cabi_multi_level_envelopes_without_gpu():
    lib = _engine.load_library()
    ref = ctypes.byref

    def fwd_ok(dtype_id, mode, flen, shape):
        d1, d2 = _analysis_pair_descs(dtype_id, mode, flen, 4, shape)
        return lib.mifwt_dwt2_fwd_pair_supported(ref(d1), ref(d2))

    assert fwd_ok(0, "reflect", 8, (1024, 1024)) == 1 and fwd_ok(0, "symmetric", 2, (256, 300)) == 1
    assert fwd_ok(0, "periodic", 8, (1024, 1024)) == 0  # level 2 would need the far side of the plane
    assert fwd_ok(0, "reflect", 16, (1024, 1024)) == 0    # long filters: per level
    assert fwd_ok(1, "reflect", 8, (1024, 1024)) == 0     # f64: per level (tile kernels)
    assert fwd_ok(0, "reflect", 8, (1024, 100)) == 0      # level-1 plane narrower than a strip
    d1, d2 = _analysis_pair_descs(0, "reflect", 8, 4, (1024, 1024))
    d2.sig_extent[0] += 1  # the second level must start from the first level's approximation
    assert lib.mifwt_dwt2_fwd_pair_supported(ref(d1), ref(d2)) == 0
    _engine.set_option(_engine.OPT_PAIR_MODE, 2)
    try:
        assert fwd_ok(0, "reflect", 8, (1024, 1024)) == 0
    finally:
        _engine.set_option(_engine.OPT_PAIR_MODE, 0)
    # null pointers are refused before anything is launched
    d1, d2 = _analysis_pair_descs(0, "reflect", 8, 4, (1024, 1024))
    assert lib.mifwt_dwt2_fwd_pair(ref(d1), ref(d2), None, None, None, None, None, None, None) == -1

    # synthesis pair: d2 = coarser level (its output = the finer level's coefficient extents), d1 = finer level
    def inv_descs(flen, out, t=(0, 0)):
        m1 = [(n + flen - 2 + tt) // 2 for n, tt in zip(out, t)]           # out = 2 m1 - L + 2 - t
        m2 = [(n + flen - 2) // 2 + (n + flen - 2) % 2 for n in m1]         # m1 = 2 m2 - L + 2 - t'
        return _dense_desc(0, "zero", flen, 2, m1, m2), _dense_desc(0, "zero", flen, 2, out, m1)

    d2, d1 = inv_descs(8, (1024, 1024))
    assert lib.mifwt_dwt2_inv_pair_supported(ref(d2), ref(d1)) == 1
    d2s, d1s = inv_descs(8, (48, 1024))
    assert lib.mifwt_dwt2_inv_pair_supported(ref(d2s), ref(d1s)) == 0  # fewer than two tiles of output rows
    d2l, d1l = inv_descs(16, (1024, 1024))
    assert lib.mifwt_dwt2_inv_pair_supported(ref(d2l), ref(d1l)) == 0
    assert lib.mifwt_dwt2_inv_pair(ref(d2), ref(d1), None, None, None, None, None, None, None) == -1

    # fused 1-D levels
    assert lib.mifwt_dwt1_fwd_tail_max_n(0) == 16384 and lib.mifwt_dwt1_fwd_tail_max_n(1) == 8192
    assert lib.mifwt_dwt1_fwd_tail(0, 8, 2, 4, 4096, 5, None, 4096, None, 0, None, None, None, None, None) == -1
    assert lib.mifwt_dwt1_inv_tail(0, 8, 4, 100, 3, None, 100, None, None, None, None, 0, None, None, None) == -1


def test_cabi_pyramid_routes_without_gpu():
    """mifwt_dwt2_fwd_pyramid_supported is host arithmetic: which of the two multi-level 2-D analysis kernels serves a call
    (0 none, 1 the streaming three-level kernel, 2 the small-plane whole-pyramid kernel)."""
    lib = _engine.load_library()

    def route(mode, flen, batch, shape, nlev, dtype_id=0):
        descs, cur = [], list(shape)
        for _ in range(nlev):
            nxt = [(n + flen - 1) // 2 for n in cur]
            descs.append(_dense_desc(dtype_id, mode, flen, batch, cur, nxt))
            cur = nxt
        refs = (ctypes.POINTER(type(descs[0])) * nlev)(*[ctypes.pointer(d) for d in descs])
        return lib.mifwt_dwt2_fwd_pyramid_supported(nlev, refs)

    assert route("reflect", 8, 64, (1024, 1024), 3) == 1
    assert route("reflect", 8, 64, (1024, 1024), 4) == 0            # the streaming kernel fuses three levels
    assert route("periodic", 8, 64, (1024, 1024), 3) == 0
    assert route("reflect", 8, 64, (4096, 4096), 3) == 0            # column groups: per level in auto mode
    assert route("reflect", 4, 4096, (64, 64), 3) == 2
    assert route("periodic", 4, 4096, (64, 64), 5) == 2             # every mode, deeper than three levels
    assert route("zero", 20, 100, (61, 47), 8) == 2                 # odd extents, 20 taps, eight levels
    assert route("reflect", 4, 4096, (64, 64), 9) == 0
    assert route("reflect", 4, 4096, (64, 64), 3, dtype_id=1) == 0  # f64: per level
    assert route("reflect", 22, 100, (61, 47), 2) == 0
    assert route("reflect", 8, 1024, (128, 128), 3) == 2            # fills a CU's LDS alone: only for big batches
    assert route("reflect", 8, 256, (128, 128), 3) == 0
    assert route("reflect", 20, 1024, (128, 128), 3) == 0           # 182 KB of LDS images
    assert route("reflect", 8, 64, (256, 256), 3) == 0
    _engine.set_option(_engine.OPT_PYRAMID_MODE, 3)
    try:
        assert route("reflect", 8, 256, (128, 128), 3) == 2
    finally:
        _engine.set_option(_engine.OPT_PYRAMID_MODE, 2)
    try:
        assert route("reflect", 4, 4096, (64, 64), 3) == 0 and route("reflect", 8, 64, (1024, 1024), 3) == 0
    finally:
        _engine.set_option(_engine.OPT_PYRAMID_MODE, 0)
    assert lib.mifwt_dwt2_fwd_pyramid(9, None, None, None, None, None, None, None) == -1


def test_cabi_inverse_pyramid_envelope_without_gpu():
    """mifwt_dwt2_inv_pyramid_supported is host arithmetic: which reconstructions the small-plane launch serves."""
    lib = _engine.load_library()

    def ok(flen, batch, out, nlev, dtype_id=0, crop=0, pitch_pad=0):
        if crop:  # top-down: every level's output is cropped by `crop` before it becomes the next level's approximation
            exts = [[10, 12]]
            for _ in range(nlev):
                exts.append([2 * m - flen + 2 - crop for m in exts[-1]])
            exts[-1] = [n + crop for n in exts[-1]]  # (the last output is not cropped)
        else:      # coefficient extents from the finest output upwards, as wavedec2 would have produced them
            exts = [list(out)]
            for _ in range(nlev):
                exts.append([(n + flen - 1) // 2 for n in exts[-1]])
            exts = exts[::-1]  # coarsest coefficient extents first, the output last
        descs = []
        for l in range(nlev):
            d = _dense_desc(dtype_id, "zero", flen, batch, exts[l + 1], exts[l])
            d.detail_stride[1] += pitch_pad
            descs.append(d)
        refs = (ctypes.POINTER(type(descs[0])) * nlev)(*[ctypes.pointer(d) for d in descs])
        return lib.mifwt_dwt2_inv_pyramid_supported(nlev, refs)

    assert ok(4, 4096, (64, 64), 3) == 1
    assert ok(20, 100, (61, 47), 2) == 1 and ok(4, 4096, (64, 64), 8) == 1
    assert ok(4, 4096, (64, 64), 3, crop=1) == 1              # the separable containers' crop of the running approximation
    assert ok(4, 4096, (64, 64), 9) == 0
    assert ok(4, 4096, (64, 64), 3, dtype_id=1) == 0          # f64: per level
    assert ok(4, 4096, (64, 64), 3, pitch_pad=2) == 0         # padded rows: per level
    assert ok(8, 1024, (128, 128), 3) == 1 and ok(8, 256, (128, 128), 3) == 0   # fills a CU's LDS alone: only for big batches
    assert ok(8, 64, (256, 256), 3) == 0
    _engine.set_option(_engine.OPT_PYRAMID_MODE, 3)
    try:
        assert ok(8, 256, (128, 128), 3) == 1
    finally:
        _engine.set_option(_engine.OPT_PYRAMID_MODE, 2)
    try:
        assert ok(4, 4096, (64, 64), 3) == 0
    finally:
        _engine.set_option(_engine.OPT_PYRAMID_MODE, 0)
    assert lib.mifwt_dwt2_inv_pyramid(9, None, None, None, None, None, None, None) == -1


def test_dwt1_long_plan_and_argument_checks():
    """mifwt_dwt1_fwd_long_levels is host arithmetic (how many levels the chunked 1-D launch fuses); mifwt_dwt1_fwd_long rejects
    bad arguments and level counts it would not fuse before touching the device."""
    import ctypes

    from ptwt_amd import _engine

    lib = _engine.load_library()
    lv = lib.mifwt_dwt1_fwd_long_levels
    F32, F64, PER, REFL = 0, 1, 3, 2
    assert lv(F32, 10, PER, 32, 1000000, 10) == 6     # the reference's speed test: down to 15 633 samples, the tail's range
    assert lv(F32, 10, PER, 32, 15633, 4) == 4        # few rows of medium length: chunked too (about one workgroup per CU)
    assert lv(F32, 10, PER, 32, 15633, 2) == 2
    assert lv(F32, 10, PER, 500, 15633, 4) == 4       # many rows as well (the one-workgroup-per-row launch is a latency chain)
    assert lv(F32, 8, REFL, 4096, 4096, 6) == 6       # the two end pieces are the whole row
    assert lv(F32, 10, PER, 4, 3000, 4) == 0          # short rows
    assert lv(F64, 10, PER, 32, 1000000, 10) == 0     # f32 only
    assert lv(F32, 22, PER, 32, 1000000, 10) == 0     # even filt_len <= 20
    assert lv(F32, 10, REFL, 32, 1000000, 1) == 0     # a single level is the per-level kernels' job
    assert lv(F32, 2, REFL, 1, 32000000, 12) == 8     # haar: no halo, the level cap
    assert lv(F32, 20, REFL, 8, 1000000, 10) == 5     # long filters: the halo rule stops earlier
    null = ctypes.c_void_p(0)
    taps = (ctypes.c_double * 10)(*([0.1] * 10))
    det = (ctypes.c_void_p * 6)(*([1] * 6))
    rs = (ctypes.c_int64 * 6)(*([0] * 6))
    one = ctypes.c_void_p(16)
    assert lib.mifwt_dwt1_fwd_long(F32, 10, PER, 32, 1000000, 6, null, 0, one, 0, det, rs, taps, taps, null) == -1    # BADARG
    det8 = (ctypes.c_void_p * 8)(*([1] * 8))
    rs8 = (ctypes.c_int64 * 8)(*([0] * 8))
    assert lib.mifwt_dwt1_fwd_long(F32, 10, PER, 32, 1000000, 7, one, 1000000, one, 0, det8, rs8, taps, taps, null) == -2  # more than one launch fuses
    assert lib.mifwt_dwt1_fwd_long(F64, 10, PER, 32, 1000000, 6, one, 1000000, one, 0, det, rs, taps, taps, null) == -2


def test_dwt1_inv_long_plan_and_argument_checks():
    """mifwt_dwt1_inv_long_supported is host arithmetic (which level counts the chunked 1-D synthesis launch fuses); the call rejects
    bad arguments before touching the device."""
    import ctypes

    from ptwt_amd import _engine

    lib = _engine.load_library()
    sup = lib.mifwt_dwt1_inv_long_supported

    def lens(n, flen, level):
        out = [n]
        for _ in range(level):
            out.append((out[-1] + flen - 1) // 2)
        return out[::-1]  # coarsest first; m[s + 1] = 2 m[s] - L + 2 - t

    m = lens(1000000, 10, 10)
    arr = lambda v: (ctypes.c_int32 * len(v))(*v)  # noqa: E731
    assert sup(0, 10, 32, 7, arr(m[3:])) == 1            # the finest seven levels of the speed-test shape
    assert sup(0, 10, 32, 8, arr(m[2:])) == 0            # eight: the halo would exceed a twelfth of a chunk
    assert sup(0, 10, 32, 3, arr(m[:4])) == 1            # 985 -> 7821 samples, 32 rows: chunked too (too few rows for one workgroup each)
    assert sup(0, 10, 500, 3, arr(m[:4])) == 1           # many rows as well
    assert sup(0, 10, 500, 2, arr(m[:3])) == 1           # 1962-sample outputs: still served (>= 1024)
    assert sup(0, 10, 500, 2, arr([250, 492, 976])) == 0  # shorter: mifwt_dwt1_inv_tail
    assert sup(1, 10, 32, 7, arr(m[3:])) == 0            # f64: half the elements per chunk, the halo rule allows six levels
    assert sup(1, 10, 32, 6, arr(m[4:])) == 1
    assert sup(0, 10, 32, 1, arr(m[9:])) == 0            # a single level is the per-level kernels' job
    bad = list(m[3:])
    bad[3] += 2
    assert sup(0, 10, 32, 7, arr(bad)) == 0              # lengths that are not a synthesis chain
    null = ctypes.c_void_p(0)
    one = ctypes.c_void_p(16)
    taps = (ctypes.c_double * 10)(*([0.1] * 10))
    det = (ctypes.c_void_p * 7)(*([16] * 7))
    rs = (ctypes.c_int64 * 7)(*([0] * 7))
    assert lib.mifwt_dwt1_inv_long(0, 10, 32, 7, arr(m[3:]), null, 0, det, rs, one, 0, taps, taps, null) == -1
    assert lib.mifwt_dwt1_inv_long(0, 10, 32, 8, arr(m[2:]), one, 0, det, rs, one, 0, taps, taps, null) == -2


# ------------------------------------------------------------------------------------------ learnable filter banks, second order (no GPU)
def test_second_order_gradients_with_learnable_taps_host_algebra(oracle_engine):
    """The autograd algebra of a double backward through a learnable filter bank (`_fwt._AnalysisLevelGrad` / `_SynthesisLevelGrad`:
    first-order gradients as ops whose backward differentiates the level, rebuilt from per-axis ops that are closed under
    differentiation, at detached copies of the inputs) against the reference's own double backward
    (tests/golden/ptwt_ref_tapgrads2.npz), with the level kernels replaced by the oracle stand-in: every mixed term data x taps,
    taps x taps, upstream gradient x taps, multi-level chains included.  The same cases run on the HIP kernels in
    tests/test_gpu_autograd.py."""
    import json

    from ptwt_amd import WaveletTensorTuple

    def weight(t, i, f=0.37):
        return torch.cos(f * torch.arange(t.numel(), dtype=torch.float64) + i).reshape(t.shape).to(t.dtype)

    def flat(coeffs):
        return [t for _, t in G.flatten_coeffs(coeffs)]

    def rebuild(coeffs, leaves):
        it = iter(leaves)
        out = [next(it)]
        for c in coeffs[1:]:
            out.append(next(it) if isinstance(c, torch.Tensor) else ({k: next(it) for k in c} if isinstance(c, dict) else type(c)(*[next(it) for _ in c])))
        return out if isinstance(coeffs, list) else tuple(out)

    z, idx = G.load("ptwt_ref_tapgrads2.npz")
    with open(os.path.join(G.GOLDEN, "pywt_filter_banks.json")) as f:
        banks = json.load(f)
    for case in idx:
        if case["fn"] in ("wavedec3", "fswavedec3") and case["shape"][1] > 12:
            continue  # (the numpy stand-in's adjoints are loops: keep the CPU tier quick)
        if case["fn"] == "swt":
            kw = {a: v for a, v in case["kw"].items()}
        k = case["key"]
        kw = {a: (tuple(v) if isinstance(v, list) else v) for a, v in case["kw"].items()}
        x = torch.from_numpy(z[k + "_x"]).requires_grad_(True)
        taps = [torch.tensor(banks[case["wavelet"]][f], dtype=torch.float64, requires_grad=True) for f in ("dec_lo", "dec_hi", "rec_lo", "rec_hi")]
        wt = WaveletTensorTuple(*taps)
        fl = flat(getattr(ptwt_amd, case["fn"])(x, wt, **kw))
        f = sum((weight(t, i) * t.square()).sum() for i, t in enumerate(fl)) / 2
        g_x, t_lo, t_hi = torch.autograd.grad(f, [x, taps[0], taps[1]], create_graph=True)
        s1 = (g_x * weight(g_x, 1, 0.53)).sum() + (t_lo * weight(t_lo, 2, 0.53)).sum() + (t_hi * weight(t_hi, 3, 0.53)).sum()
        for got, nme in zip(torch.autograd.grad(s1, [x, taps[0], taps[1]]), ("a_dx", "a_dlo", "a_dhi")):
            assert G.relerr(got.numpy(), z["%s_%s" % (k, nme)]) < 1e-11, (case, nme)
        coeffs = getattr(ptwt_amd, case["fn"])(x.detach(), case["wavelet"], **kw)
        leaves = [t.detach().clone().requires_grad_(True) for t in flat(coeffs)]
        rkw = {a: v for a, v in kw.items() if a in ("axis", "axes")}
        y = getattr(ptwt_amd, case["rec"])(rebuild(coeffs, leaves), wt, **rkw)
        grads = torch.autograd.grad((weight(y, 7) * y.square()).sum() / 2, leaves + [taps[2], taps[3]], create_graph=True)
        s2 = sum((gc * weight(gc, 4 + i, 0.53)).sum() for i, gc in enumerate(grads[:-2]))
        s2 = s2 + (grads[-2] * weight(grads[-2], 2, 0.53)).sum() + (grads[-1] * weight(grads[-1], 3, 0.53)).sum()
        d2 = torch.autograd.grad(s2, leaves + [taps[2], taps[3]])
        for i, got in enumerate(d2[:-2]):
            assert G.relerr(got.numpy(), z["%s_s_dc%d" % (k, i)]) < 1e-11, (case, "s_dc", i)
        assert G.relerr(d2[-2].numpy(), z[k + "_s_dlo"]) < 1e-11 and G.relerr(d2[-1].numpy(), z[k + "_s_dhi"]) < 1e-11, case


def test_differentiable_calls_take_the_multi_level_ops_host_side(oracle_engine):
    """Host logic of the differentiable multi-level ops (`_fwt._AnalysisPyramid`, `_AnalysisTail`, `_SynthesisPyramid`,
    `_SynthesisChain1d`) with the oracle stand-in: same values as the plain calls, gradients equal to those of the per-level ops
    (autograd through the stand-in's adjoints), containers unchanged."""
    torch.manual_seed(2)
    for fn, rec, shape, kw in (("wavedec2", "waverec2", (2, 40, 44), dict(level=3, mode="symmetric")), ("wavedec", "waverec", (3, 60), dict(level=3, mode="reflect")),
                               ("fswavedec2", "fswaverec2", (2, 33, 41), dict(level=2, mode="zero"))):
        x = torch.randn(*shape, dtype=torch.float32, requires_grad=True)
        with torch.no_grad():
            plain = getattr(ptwt_amd, fn)(x, "db2", **kw)
        coeffs = getattr(ptwt_amd, fn)(x, "db2", **kw)
        fl, pl = [t for _, t in G.flatten_coeffs(coeffs)], [t for _, t in G.flatten_coeffs(plain)]
        assert type(coeffs) is type(plain) and all(a.requires_grad and torch.equal(a.detach(), b) for a, b in zip(fl, pl))
        ws = [torch.cos(torch.arange(t.numel(), dtype=torch.float32) * 0.3).reshape(t.shape) for t in fl]
        (gx,) = torch.autograd.grad(sum((w * t).sum() for w, t in zip(ws, fl)), x)
        # the adjoint identity against the plain forward: <A x, w> = <x, A^T w>
        lhs = sum((w * t).sum() for w, t in zip(ws, pl)).item()
        assert abs(lhs - (gx * x.detach()).sum().item()) < 1e-3 * max(1.0, abs(lhs))
        leaves = [t.detach().clone().requires_grad_(True) for t in pl]
        it = iter(leaves)
        rebuilt = [next(it)] + [({k: next(it) for k in c} if isinstance(c, dict) else (type(c)(*[next(it) for _ in c]) if isinstance(c, tuple) else next(it))) for c in plain[1:]]
        y = getattr(ptwt_amd, rec)(rebuilt if isinstance(plain, list) else tuple(rebuilt), "db2")
        v = torch.sin(torch.arange(y.numel(), dtype=torch.float32) * 0.2).reshape(y.shape)
        gl = torch.autograd.grad((v * y).sum(), leaves)
        lhs = (v * y.detach()).sum().item()
        assert abs(lhs - sum((g * t.detach()).sum().item() for g, t in zip(gl, leaves))) < 1e-3 * max(1.0, abs(lhs))


def test_every_kernel_id_is_documented_in_the_header():
    """`enum KernelId` (csrc/mifwt_common.h) against the kernel-id list in the comment of `mifwt_kernel_id` (include/mifwt.h): a new
    kernel family must appear in the public header."""
    import os
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "pytorch-wavelet-toolbox_amd", "csrc", "mifwt_common.h")).read()
    body = re.search(r"enum KernelId \{(.*?)\};", src, flags=re.S).group(1)
    ids = sorted({int(v) for v in re.findall(r"=\s*(\d+)", body)})
    assert ids[0] == 0 and 24 in ids and 25 in ids
    hdr = open(os.path.join(root, "include", "mifwt.h")).read()
    block = hdr[hdr.index("generic per-axis passes (any strides"):hdr.index("int mifwt_kernel_id(")]
    documented = {int(v) for head in re.findall(r"^ \*\s{3}((?:\d+ / )?\d+)\s+\S", block, flags=re.M) for v in head.split(" / ")}
    documented.add(0)
    assert set(ids) <= documented, sorted(set(ids) - documented)


def test_half_storage_is_scoped_and_thread_local():
    """fp16 storage is an engine extension the reference refuses (src/ptwt/constants.py:27): `with ptwt_amd.half_storage():` switches it
    on for the calls of this thread / task only; `set_half_storage` is the process-wide default it falls back to."""
    import threading

    import torch

    import ptwt_amd
    from ptwt_amd import constants as C

    assert torch.float16 not in C.supported_dtypes()
    seen = {}
    with ptwt_amd.half_storage():
        assert torch.float16 in C.supported_dtypes()
        t = threading.Thread(target=lambda: seen.setdefault("other", torch.float16 in C.supported_dtypes()))
        t.start()
        t.join()
        with ptwt_amd.half_storage(False):
            assert torch.float16 not in C.supported_dtypes()
        assert torch.float16 in C.supported_dtypes()
    assert seen["other"] is False
    assert torch.float16 not in C.supported_dtypes()
    ptwt_amd.set_half_storage(True)
    try:
        assert torch.float16 in C.supported_dtypes()
        with ptwt_amd.half_storage(False):
            assert torch.float16 not in C.supported_dtypes()
    finally:
        ptwt_amd.set_half_storage(False)


def test_f64_volume_routes_without_a_gpu():
    """`mifwt_kernel_id` answers on the host: f64 volumes take the f64 instances of the depth-walking kernels (ids 24 / 25) from the
    measured thresholds on (per filter length, csrc/mifwt_api.hip; profiles/r05w_f64_walk_vs_planes.txt), the composed route (5 / 6)
    below them, for ten taps, for rows of more than 256 samples and where a row group's coefficient pieces exceed 5 KiB."""
    import torch
    from ptwt_amd import _engine

    kid = _engine.kernel_id
    f64 = torch.float64
    assert kid(3, f64, "zero", 4, 8, (256, 256, 256)) == 24 and kid(3, f64, "zero", 4, 8, (256, 256, 256), direction=1) == 25
    assert kid(3, f64, "periodic", 4, 2, (33, 34, 35)) == 24 and kid(3, f64, "zero", 4, 2, (33, 34, 35), direction=1) == 6
    assert kid(3, f64, "zero", 4, 2, (41, 42, 43), direction=1) == 25
    assert kid(3, f64, "zero", 4, 2, (20, 21, 22)) == 5 and kid(3, f64, "zero", 4, 2, (20, 21, 22), direction=1) == 6
    assert kid(3, f64, "zero", 4, 2, (40, 40, 300)) == 5  # rows of more than 256 doubles
    assert kid(3, f64, "reflect", 6, 2, (66, 66, 66)) == 24 and kid(3, f64, "reflect", 6, 2, (40, 40, 40)) == 5
    assert kid(3, f64, "reflect", 8, 2, (100, 100, 100)) == 24 and kid(3, f64, "reflect", 8, 2, (66, 66, 66)) == 5
    assert kid(3, f64, "zero", 10, 2, (128, 128, 128)) == 5 and kid(3, f64, "zero", 10, 2, (128, 128, 128), direction=1) == 6
    assert kid(3, f64, "zero", 8, 2, (100, 100, 100), direction=1) == 25 and kid(3, f64, "zero", 8, 2, (256, 256, 256), direction=1) == 6
    # f32 keeps its routes: walk from 2^22 samples on, bricks below, eight taps from 2^20 on
    f32 = torch.float32
    assert kid(3, f32, "zero", 4, 8, (256, 256, 256)) == 24 and kid(3, f32, "zero", 4, 4, (129, 129, 129)) == 9
    assert kid(3, f32, "zero", 8, 8, (128, 128, 128)) == 24 and kid(3, f32, "zero", 8, 8, (54, 54, 54)) == 5


def test_3d_analysis_routes_of_round_6_without_a_gpu():
    """Batch-aware 3-D analysis routing (csrc/mifwt_api.hip, `dwt3_fwd_slab_pays`; profiles/r06k / r06p / r06y_walk3_routes.txt): the
    depth-walking kernel (id 24) takes smaller volumes once the batch fills the chip, and eight / ten taps on rows of at most 128 samples
    take its slab form where the measured table says so — else the bricks (9) / the composed route (5)."""
    import torch
    from ptwt_amd import _engine

    kid, f32 = _engine.kernel_id, torch.float32
    # L <= 6: from 2^22 samples a volume; from 2^21 with eight volumes; from 10^5 with sixteen
    assert kid(3, f32, "zero", 4, 8, (129, 129, 129)) == 24 and kid(3, f32, "zero", 4, 4, (129, 129, 129)) == 9
    assert kid(3, f32, "zero", 4, 8, (66, 66, 66)) == 9 and kid(3, f32, "periodic", 4, 32, (51, 51, 51)) == 24
    assert kid(3, f32, "periodic", 6, 32, (52, 52, 52)) == 24 and kid(3, f32, "periodic", 4, 32, (27, 27, 27)) == 9
    # ten taps (the reference's 3-D speed shape: 32 x 100^3 db5 periodic): slab form from 2^23 samples a batch on big volumes, 2.4 M on small ones
    assert kid(3, f32, "periodic", 10, 32, (100, 100, 100)) == 24 and kid(3, f32, "periodic", 10, 16, (100, 100, 100)) == 24
    assert kid(3, f32, "periodic", 10, 8, (100, 100, 100)) == 5 and kid(3, f32, "periodic", 10, 1, (100, 100, 100)) == 5
    assert kid(3, f32, "periodic", 10, 32, (54, 54, 54)) == 24 and kid(3, f32, "periodic", 10, 8, (54, 54, 54)) == 5
    assert kid(3, f32, "periodic", 10, 32, (31, 31, 31)) == 5  # below 10^5 samples a volume: composed
    assert kid(3, f32, "reflect", 10, 32, (100, 100, 129)) == 5  # rows of more than 128 samples are not the slab form's
    # eight taps: slab form from 4 M samples a batch (2.3 M on small volumes); big volumes keep the walk kernel either way
    assert kid(3, f32, "symmetric", 8, 4, (100, 100, 100)) == 24 and kid(3, f32, "symmetric", 8, 2, (100, 100, 100)) == 5
    assert kid(3, f32, "symmetric", 8, 16, (53, 53, 53)) == 24 and kid(3, f32, "symmetric", 8, 8, (53, 53, 53)) == 5
    assert kid(3, f32, "zero", 8, 1, (128, 128, 128)) == 24 and kid(3, f32, "zero", 8, 32, (30, 30, 30)) == 5
    # f64 and half storage never take the slab form
    assert kid(3, torch.float64, "periodic", 10, 32, (100, 100, 100)) == 5


def test_graph_free_calls_replay_their_route(oracle_engine):
    """`_fwt._route_memo`: a decomposition that builds no graph replays the launches its geometry took last time — same results, no
    memo for calls that raise, none for differentiable calls, cleared when a routing option changes."""
    import numpy as np
    import torch
    import ptwt_amd
    from ptwt_amd import _engine, _fwt

    _fwt._route_memo.clear()
    x = torch.from_numpy(np.random.default_rng(5).standard_normal((2, 40, 52)))
    first = ptwt_amd.wavedec2(x, "db2", mode="symmetric", level=3)
    assert len(_fwt._route_memo) == 1
    (steps,) = _fwt._route_memo.values()
    assert len(steps) >= 1 and all(kind in (0, 1, 2, 3) for kind, _ in steps)
    again = ptwt_amd.wavedec2(x, "db2", mode="symmetric", level=3)
    assert len(_fwt._route_memo) == 1
    for a, b in zip([first[0]] + [t for lv in first[1:] for t in lv], [again[0]] + [t for lv in again[1:] for t in lv]):
        assert torch.equal(a, b)
    # another mode / level / layout is another geometry; 1-D and 3-D calls memoise too
    ptwt_amd.wavedec2(x, "db2", mode="zero", level=2)
    ptwt_amd.wavedec2(x.transpose(1, 2), "db2", mode="symmetric", level=3)
    ptwt_amd.wavedec(x[0], "db3", level=2)
    ptwt_amd.wavedec3(x.reshape(2, 8, 5, 52), "haar", level=1)
    assert len(_fwt._route_memo) == 5
    # a geometry whose pad check raises is never memoised (the reference's error comes back every time)
    n = len(_fwt._route_memo)
    for _ in range(2):
        with pytest.raises(RuntimeError):
            ptwt_amd.wavedec2(x[:, :5, :5], "db4", mode="reflect", level=1)
    assert len(_fwt._route_memo) == n
    # differentiable calls take the autograd ops, not the replay
    xg = x.clone().requires_grad_(True)
    ptwt_amd.wavedec2(xg, "db2", mode="symmetric", level=3)
    assert len(_fwt._route_memo) == n
    with torch.no_grad():
        ptwt_amd.wavedec2(xg, "db2", mode="symmetric", level=3)  # (same geometry as the first call: replayed)
    assert len(_fwt._route_memo) == n
    # a routing option clears it with the plans
    _engine.set_option(_engine.OPT_TILE_MODE, 0)
    assert len(_fwt._route_memo) == 0


def test_slab_plan_invariants_without_a_gpu():
    """`mifwt_dwt3_fwd_slab_plan`: what the slab form of kernel 24 (csrc/mifwt_dwt3_fwd_slab.hip) relies on, for every geometry of a sweep —
    the workgroup's LDS fits a CU, a loader wave keeps at most 20 requests and 8 x 64 pad samples a slice, the row pitches spread the
    16-byte accesses of neighbouring rows over the banks (4 x odd floats, 2 x odd pairs), a row holds its pads and the row pass's last
    group of four outputs, slabs and depth segments cover the volume."""
    lib = _engine.load_library()
    lib.mifwt_dwt3_fwd_slab_plan.restype = ctypes.c_int
    lib.mifwt_dwt3_fwd_slab_plan.argtypes = [ctypes.POINTER(_engine.LevelDesc), ctypes.POINTER(ctypes.c_int), ctypes.c_int]
    out = (ctypes.c_int * 12)()

    def desc(batch, shape, flen, dtype=0, mode="periodic"):
        nd = len(shape)
        coef = tuple((n + flen - 1) // 2 for n in shape)
        d = _engine.LevelDesc()
        d.ndim, d.dtype, d.mode, d.filt_len, d.batch = nd, dtype, _engine.MODE_IDS[mode], flen, batch
        s_sig = s_coef = 1
        for a in reversed(range(nd)):
            d.sig_extent[a], d.coef_extent[a] = shape[a], coef[a]
            d.sig_stride[1 + a] = s_sig
            d.approx_stride[1 + a] = d.detail_stride[1 + a] = s_coef
            s_sig, s_coef = s_sig * shape[a], s_coef * coef[a]
        d.sig_stride[0] = s_sig
        d.approx_stride[0] = d.detail_stride[0] = s_coef << nd
        return d, coef

    def plan(batch, shape, flen, dtype=0, mode="periodic"):
        d, coef = desc(batch, shape, flen, dtype, mode)
        return lib.mifwt_dwt3_fwd_slab_plan(ctypes.byref(d), out, 12), list(out), coef

    seen = 0
    for flen in (8, 10):
        for batch in (1, 7, 32, 500):
            for shape in ((100, 100, 100), (54, 54, 54), (31, 31, 31), (flen, flen, flen), (12, 128, 128), (200, 17, 33), (13, 97, 11),
                          (64, 128, 10), (20, 31, 128), (40, 41, 127), (9 + flen, 66, 65)):
                rc, (rw, ngroups, ncw, pitch_f, rpq, rin_max, wfp, npad, nseg, seg_out, lds, pays), (Do, Ho, Wo) = plan(batch, shape, flen)
                assert rc == 12, (rc, flen, batch, shape)
                seen += 1
                W, HL = shape[2], flen - 2
                assert rw % 2 == 0 and rw * ngroups >= Ho and rw * (ngroups - 1) < Ho
                assert 1 <= ncw <= 12 and 64 * ncw >= (rw // 2) * Wo
                assert rin_max == 2 * rw + HL and npad == HL + 2 * Wo - W
                assert pitch_f % 4 == 0 and (pitch_f // 4) % 2 == 1 and pitch_f >= 8 + max(2 * Wo, W) and pitch_f <= 4 * 64
                assert rpq == 64 // (pitch_f // 4) >= 1
                assert wfp % 2 == 0 and (wfp // 2) % 2 == 1 and wfp >= 4 * ((Wo + 3) // 4)
                nreq = -(-rin_max // rpq)
                nown = -(-nreq // 4)
                assert nown <= 20 and nown * rpq * npad <= 64 * 8
                raw = (-(-rin_max // rpq) * rpq) * pitch_f * 4 + 32
                assert lds == 3 * raw + rin_max * wfp * 8 and lds <= 160 * 1024
                assert nseg >= 1 and seg_out >= 1 and nseg * seg_out >= Do and (nseg - 1) * seg_out < Do
                assert pays in (0, 1)
    assert seen == 2 * 4 * 11
    # where the form does not apply: other filter lengths, rows of more than 128 samples, an extent shorter than the filter, f64 / f16, two axes
    assert plan(32, (100, 100, 100), 6)[0] == -2 and plan(32, (100, 100, 129), 10)[0] == -2 and plan(32, (9, 100, 100), 10)[0] == -2
    assert plan(32, (100, 100, 100), 10, dtype=1)[0] == -2
    assert plan(4, (100, 100), 10)[0] == -2
    d, _ = desc(32, (100, 100, 100), 10)
    assert lib.mifwt_dwt3_fwd_slab_plan(ctypes.byref(d), out, 11) == -1 and lib.mifwt_dwt3_fwd_slab_plan(None, out, 12) == -1
    # the reference's 3-D speed shape: two slabs of 28 rows, twelve compute waves, rows of 116 floats two per request, the default route
    assert plan(32, (100, 100, 100), 10)[1][:8] == [28, 2, 12, 116, 2, 64, 58, 16] and plan(32, (100, 100, 100), 10)[1][11] == 1
    assert plan(2, (100, 100, 100), 10)[1][11] == 0
