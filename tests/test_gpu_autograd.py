"""GPU tests (``-m gpu``) of reverse-mode differentiation through the ten entry points.

Oracle: gradients computed by the REFERENCE's own autograd (ATen conv backward), committed as
tests/golden/ptwt_ref_grads.npz by tests/golden/make_ptwt_ref_grad_goldens.py; plus adjoint identities
<A x, w> = <x, A^T w> at sizes that exercise the fused kernels.  Tolerances: fp64 1e-11 norm-wise vs the reference,
fp32 identities 2e-5 relative (sums of ~1e6 products)."""
import numpy as np
import pytest
import torch

import ptwt_amd
from ptwt_amd import _engine
from tests import _golden as G

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def weight(t, i):
    return torch.cos(0.37 * torch.arange(t.numel(), dtype=torch.float64, device=t.device) + i).reshape(t.shape).to(t.dtype)


def flat(coeffs):
    return [t for _, t in G.flatten_coeffs(coeffs)]


def rebuild(coeffs, leaves):
    it = iter(leaves)
    out = [next(it)]
    for c in coeffs[1:]:
        if isinstance(c, torch.Tensor):
            out.append(next(it))
        elif isinstance(c, dict):
            out.append({k: next(it) for k in c})
        else:
            out.append(type(c)(*[next(it) for _ in c]))
    return out if isinstance(coeffs, list) else tuple(out)


def test_gradients_vs_reference_autograd():
    z, idx = G.load("ptwt_ref_grads.npz")
    for case in idx:
        k = case["key"]
        kw = {a: (tuple(v) if isinstance(v, list) else v) for a, v in case["kw"].items()}
        x = torch.from_numpy(z[k + "_x"]).to(dev()).requires_grad_(True)
        coeffs = getattr(ptwt_amd, case["fn"])(x, case["wavelet"], **kw)
        fl = flat(coeffs)
        assert len(fl) == case["ncoef"]
        loss = sum((weight(t, i) * t).sum() for i, t in enumerate(fl))
        (gx,) = torch.autograd.grad(loss, x)
        assert G.relerr(gx.cpu().numpy(), z[k + "_gx"]) < 1e-11, (case, "analysis backward")
        leaves = [t.detach().clone().requires_grad_(True) for t in fl]
        rkw = {a: v for a, v in kw.items() if a in ("axis", "axes")}
        y = getattr(ptwt_amd, case["rec"])(rebuild(coeffs, leaves), case["wavelet"], **rkw)
        gl = torch.autograd.grad((weight(y, 7) * y).sum(), leaves)
        for i, g in enumerate(gl):
            assert G.relerr(g.cpu().numpy(), z["%s_gc%d" % (k, i)]) < 1e-11, (case, "synthesis backward", i)


@pytest.mark.parametrize("mode", ["zero", "reflect", "periodic", "symmetric", "constant"])
@pytest.mark.parametrize("fn,rec,shape", [("wavedec", "waverec", (3, 5001)), ("wavedec2", "waverec2", (3, 203, 610)),
                                          ("wavedec3", "waverec3", (2, 37, 41, 45))])
def test_adjoint_identities_fp32(mode, fn, rec, shape):
    """<A x, w> == <x, A^T w> and <S c, v> == <c, S^T v> in fp32 at sizes where the forward (and, for zero mode,
    the backward) runs on the fused / streaming kernels."""
    torch.manual_seed(7)
    x = torch.randn(*shape, device=dev(), requires_grad=True)
    coeffs = getattr(ptwt_amd, fn)(x, "db4", mode=mode, level=2)
    fl = flat(coeffs)
    ws = [torch.randn_like(t) for t in fl]
    lhs = sum((w * t).sum() for w, t in zip(ws, fl))
    (gx,) = torch.autograd.grad(lhs, x)
    rhs = (gx.double() * x.detach().double()).sum()
    assert abs(lhs.item() - rhs.item()) <= 2e-5 * max(1.0, abs(rhs.item())) + 2e-2, (lhs.item(), rhs.item())
    leaves = [t.detach().clone().requires_grad_(True) for t in fl]
    y = getattr(ptwt_amd, rec)(rebuild(coeffs, leaves), "db4")
    v = torch.randn_like(y)
    lhs = (v * y).sum()
    gl = torch.autograd.grad(lhs, leaves)
    rhs = sum((g.double() * t.detach().double()).sum() for g, t in zip(gl, leaves))
    assert abs(lhs.item() - rhs.item()) <= 2e-5 * max(1.0, abs(rhs.item())) + 2e-2, (lhs.item(), rhs.item())


@pytest.mark.parametrize("mode", ["zero", "reflect", "symmetric", "constant"])
@pytest.mark.parametrize("shape,wavelet,level,pmode", [((2, 300, 520), "db4", 3, 1), ((3, 203, 610), "db2", 3, 1), ((6, 96, 80), "db3", 3, 3),
                                                        ((2, 512, 512), "haar", 5, 0)])
def test_fused_forward_of_differentiable_calls(mode, shape, wavelet, level, pmode):
    """A `wavedec2` that asks for gradients w.r.t. the data runs its forward on the multi-level launches (kernel ids 16 / 20:
    `_fwt._AnalysisPyramid`; src/ptwt/conv_transform_2.py:142-149), not level by level: same kernel ids as the plain call, coefficients
    bit-identical to it; gradients equal to those of the per-level ops (pyramid launches switched off) within fp32 rounding, the
    adjoint identity, and a Hessian-vector product through the fused op (its backward is made of differentiable level ops)."""
    torch.manual_seed(9)
    x = torch.randn(*shape, device=dev(), requires_grad=True)
    _engine.set_option(_engine.OPT_PYRAMID_MODE, pmode)
    try:
        _engine.level_events = []
        coeffs = ptwt_amd.wavedec2(x, wavelet, mode=mode, level=level)
        torch.cuda.synchronize()
        kids = [e[1] for e in _engine.level_events]
        _engine.level_events = None
        assert kids[0] in (_engine.KID_PYRAMID, _engine.KID_SMALL), kids
        with torch.no_grad():
            _engine.level_events = []
            plain = ptwt_amd.wavedec2(x, wavelet, mode=mode, level=level)
            torch.cuda.synchronize()
            assert [e[1] for e in _engine.level_events][0] == kids[0]  # (the levels after the fused ones may pair up in the plain call)
            _engine.level_events = None
        fl = flat(coeffs)
        for a, b in zip(fl, flat(plain)):
            assert a.requires_grad and torch.equal(a.detach(), b)
        ws = [torch.randn_like(t) for t in fl]
        lhs = sum((w * t).sum() for w, t in zip(ws, fl))
        (gx,) = torch.autograd.grad(lhs, x, retain_graph=True)
        rhs = (gx.double() * x.detach().double()).sum()
        assert abs(lhs.item() - rhs.item()) <= 2e-5 * max(1.0, abs(rhs.item())) + 2e-2, (lhs.item(), rhs.item())
        # Hessian-vector product of f = sum w c^2 / 2 through the fused op against the first-order gradient of f at v
        v = torch.randn_like(x)

        def f(t):
            return sum((w * c.square()).sum() for w, c in zip(ws, flat(ptwt_amd.wavedec2(t, wavelet, mode=mode, level=level)))) / 2

        (g,) = torch.autograd.grad(f(x), x, create_graph=True)
        (hv,) = torch.autograd.grad((g * v).sum(), x)
        vv = v.clone().requires_grad_(True)
        (want_hv,) = torch.autograd.grad(f(vv), vv)
        assert G.relerr(hv.cpu().numpy(), want_hv.cpu().numpy()) < 2e-6
    finally:
        _engine.level_events = None
        _engine.set_option(_engine.OPT_PYRAMID_MODE, 0)
    # the per-level ops (every multi-level launch off)
    _engine.set_option(_engine.OPT_PYRAMID_MODE, 2)
    _engine.set_option(_engine.OPT_PAIR_MODE, 2)
    try:
        x2 = x.detach().clone().requires_grad_(True)
        fl2 = flat(ptwt_amd.wavedec2(x2, wavelet, mode=mode, level=level))
        (gx2,) = torch.autograd.grad(sum((w * t).sum() for w, t in zip(ws, fl2)), x2)
    finally:
        _engine.set_option(_engine.OPT_PYRAMID_MODE, 0)
        _engine.set_option(_engine.OPT_PAIR_MODE, 0)
    assert G.relerr(gx.cpu().numpy(), gx2.cpu().numpy()) < 2e-6


def test_zero_mode_backward_is_one_multi_level_synthesis_launch():
    """Zero mode: the adjoint of a multi-level analysis launch is ONE multi-level synthesis launch with the dec taps reversed (kernels 22 /
    21) when no graph of the backward is asked for; with create_graph=True the per-level (differentiable) adjoints run instead.  Same
    gradient either way (fp32 rounding)."""
    torch.manual_seed(16)
    for shape, pmode, want in (((4, 600, 520), 0, _engine.KID_INV_PYRAMID), ((8, 96, 80), 3, _engine.KID_INV_SMALL)):
        _engine.set_option(_engine.OPT_PYRAMID_MODE, pmode)
        try:
            x = torch.randn(*shape, device=dev(), requires_grad=True)
            fl = flat(ptwt_amd.wavedec2(x, "db4", mode="zero", level=3))
            ws = [torch.randn_like(t) for t in fl]
            loss = sum((w * t).sum() for w, t in zip(ws, fl))
            _engine.level_events = []
            (g1,) = torch.autograd.grad(loss, x, retain_graph=True)
            torch.cuda.synchronize()
            kids1 = [e[1] for e in _engine.level_events]
            _engine.level_events = []
            (g2,) = torch.autograd.grad(loss, x, create_graph=True)
            torch.cuda.synchronize()
            kids2 = [e[1] for e in _engine.level_events]
            _engine.level_events = None
        finally:
            _engine.level_events = None
            _engine.set_option(_engine.OPT_PYRAMID_MODE, 0)
        assert kids1 == [want], (shape, kids1)
        assert len(kids2) == 3 and want not in kids2, (shape, kids2)
        assert G.relerr(g1.cpu().numpy(), g2.detach().cpu().numpy()) < 2e-6


def test_fused_forward_gradients_at_config2_size():
    """Forward + backward of `wavedec2` db4 level 3 on 64 x 1024^2 (BASELINE config 2 with gradients; the benchmark workload
    `wavedec2_bwd_...`): one forward launch (kernel 16), gradient against the per-level ops on every image, adjoint identity."""
    torch.manual_seed(10)
    x = torch.randn(64, 1024, 1024, device=dev(), requires_grad=True)
    _engine.level_events = []
    fl = flat(ptwt_amd.wavedec2(x, "db4", level=3))
    torch.cuda.synchronize()
    kids = [e[1] for e in _engine.level_events]
    _engine.level_events = None
    assert kids == [_engine.KID_PYRAMID], kids
    ws = [torch.randn_like(t) for t in fl]
    lhs = sum((w * t).sum(dtype=torch.float64) for w, t in zip(ws, fl))
    (gx,) = torch.autograd.grad(lhs, x)
    rhs = (gx.double() * x.detach().double()).sum()
    assert abs(lhs.item() - rhs.item()) <= 2e-5 * abs(rhs.item()) + 1.0, (lhs.item(), rhs.item())
    _engine.set_option(_engine.OPT_PYRAMID_MODE, 2)
    _engine.set_option(_engine.OPT_PAIR_MODE, 2)
    try:
        x2 = x.detach().clone().requires_grad_(True)
        fl2 = flat(ptwt_amd.wavedec2(x2, "db4", level=3))
        (gx2,) = torch.autograd.grad(sum((w * t).sum(dtype=torch.float64) for w, t in zip(ws, fl2)), x2)
    finally:
        _engine.set_option(_engine.OPT_PYRAMID_MODE, 0)
        _engine.set_option(_engine.OPT_PAIR_MODE, 0)
    err = (gx - gx2).flatten(1).norm(dim=1) / gx2.flatten(1).norm(dim=1)
    assert float(err.max()) < 2e-6, float(err.max())


@pytest.mark.parametrize("fn,rec", [("wavedec2", "waverec2"), ("fswavedec2", "fswaverec2")])
@pytest.mark.parametrize("shape,wavelet,level,pmode", [((2, 600, 520), "db4", 3, 0), ((3, 403, 611), "db2", 2, 0), ((6, 95, 81), "db3", 3, 3),
                                                        ((2, 1024, 1024), "db5", 4, 0)])
def test_fused_synthesis_of_differentiable_calls(shape, wavelet, level, pmode, fn, rec):
    """A `waverec2` that asks for gradients w.r.t. the coefficients runs on the multi-level launches (kernel ids 22 / 21:
    `_fwt._SynthesisPyramid`; src/ptwt/conv_transform_2.py:222-249): same kernel ids and bit-identical output as the plain call,
    gradients of every coefficient tensor equal to those of the per-level ops within fp32 rounding, a second-order product through it."""
    torch.manual_seed(12)
    x = torch.randn(*shape, device=dev())
    coeffs = getattr(ptwt_amd, fn)(x, wavelet, mode="symmetric", level=level)
    leaves = [t.detach().clone().requires_grad_(True) for t in flat(coeffs)]
    _engine.set_option(_engine.OPT_PYRAMID_MODE, pmode)
    try:
        _engine.level_events = []
        y = getattr(ptwt_amd, rec)(rebuild(coeffs, leaves), wavelet)
        torch.cuda.synchronize()
        kids = [e[1] for e in _engine.level_events]
        _engine.level_events = []
        with torch.no_grad():
            y0 = getattr(ptwt_amd, rec)(rebuild(coeffs, leaves), wavelet)
        torch.cuda.synchronize()
        kids0 = [e[1] for e in _engine.level_events]
        _engine.level_events = None
        assert kids[-1] in (_engine.KID_INV_PYRAMID, _engine.KID_INV_SMALL) and kids[-1] == kids0[-1], (kids, kids0)
        assert y.requires_grad and torch.equal(y.detach(), y0)
        v = torch.randn_like(y)
        gl = torch.autograd.grad((v * y).sum(), leaves, create_graph=True)
        us = [torch.randn_like(t) for t in leaves]
        # d/dv of <grad_c <v, R c>, u> = R u: a second backward through the fused op's (differentiable) adjoints
        vv = v.clone().requires_grad_(True)
        gl2 = torch.autograd.grad((vv * getattr(ptwt_amd, rec)(rebuild(coeffs, leaves), wavelet)).sum(), leaves, create_graph=True)
        (ru,) = torch.autograd.grad(sum((a * b).sum() for a, b in zip(gl2, us)), vv)
        with torch.no_grad():
            want_ru = getattr(ptwt_amd, rec)(rebuild(coeffs, us), wavelet)
        assert G.relerr(ru.cpu().numpy(), want_ru.cpu().numpy()) < 2e-6
    finally:
        _engine.level_events = None
        _engine.set_option(_engine.OPT_PYRAMID_MODE, 0)
    _engine.set_option(_engine.OPT_PYRAMID_MODE, 2)
    _engine.set_option(_engine.OPT_PAIR_MODE, 2)
    try:
        leaves2 = [t.detach().clone().requires_grad_(True) for t in leaves]
        y2 = getattr(ptwt_amd, rec)(rebuild(coeffs, leaves2), wavelet)
        gl_ref = torch.autograd.grad((v * y2).sum(), leaves2)
    finally:
        _engine.set_option(_engine.OPT_PYRAMID_MODE, 0)
        _engine.set_option(_engine.OPT_PAIR_MODE, 0)
    for i, (a, b) in enumerate(zip(gl, gl_ref)):
        assert G.relerr(a.detach().cpu().numpy(), b.cpu().numpy()) < 2e-6, i
    # a plain backward (no graph of the backward wanted): the adjoint of the whole reconstruction is ONE zero-mode multi-level ANALYSIS
    # launch with the rec taps reversed (kernels 16 / 20) where the library serves the geometry — same gradients
    _engine.set_option(_engine.OPT_PYRAMID_MODE, pmode)
    try:
        leaves3 = [t.detach().clone().requires_grad_(True) for t in leaves]
        y3 = getattr(ptwt_amd, rec)(rebuild(coeffs, leaves3), wavelet)
        _engine.level_events = []
        gl3 = torch.autograd.grad((v * y3).sum(), leaves3)
        torch.cuda.synchronize()
        kids_b = [e[1] for e in _engine.level_events]
    finally:
        _engine.level_events = None
        _engine.set_option(_engine.OPT_PYRAMID_MODE, 0)
    for i, (a, b) in enumerate(zip(gl3, gl_ref)):
        assert a.shape == b.shape and G.relerr(a.cpu().numpy(), b.cpu().numpy()) < 2e-6, (i, kids_b)
    if shape in ((2, 600, 520), (6, 95, 81)) and rec == "waverec2":
        assert kids_b and kids_b[0] in (_engine.KID_PYRAMID, _engine.KID_SMALL) and len(kids_b) < level, kids_b


@pytest.mark.parametrize("mode", ["zero", "reflect", "periodic", "symmetric"])
@pytest.mark.parametrize("shape,wavelet,level", [((3, 5001), "db4", 5), ((2, 40000), "db5", 8), ((40, 1000), "db2", 4), ((1, 300000), "haar", 10)])
def test_fused_1d_launches_of_differentiable_calls(mode, shape, wavelet, level):
    """`wavedec` / `waverec` with gradients w.r.t. the data run on the multi-level 1-D launches (kernel ids 17 / 14 and 18 / 15:
    `_fwt._AnalysisTail`, `_SynthesisChain1d`; src/ptwt/conv_transform.py:133-140, :184-199): same kernel ids and bit-identical values as
    the plain calls, gradients equal to those of the per-level ops (multi-level launches off) within fp32 rounding."""
    torch.manual_seed(14)
    x = torch.randn(*shape, device=dev(), requires_grad=True)
    _engine.level_events = []
    coeffs = ptwt_amd.wavedec(x, wavelet, mode=mode, level=level)
    torch.cuda.synchronize()
    kids = [e[1] for e in _engine.level_events]
    _engine.level_events = []
    with torch.no_grad():
        plain = ptwt_amd.wavedec(x, wavelet, mode=mode, level=level)
    torch.cuda.synchronize()
    kids0 = [e[1] for e in _engine.level_events]
    _engine.level_events = None
    assert kids == kids0 and any(k in (14, 17) for k in kids), (kids, kids0)
    for a, b in zip(coeffs, plain):
        assert a.requires_grad and torch.equal(a.detach(), b)
    ws = [torch.randn_like(t) for t in coeffs]
    (gx,) = torch.autograd.grad(sum((w * t).sum() for w, t in zip(ws, coeffs)), x)
    leaves = [t.detach().clone().requires_grad_(True) for t in coeffs]
    _engine.level_events = []
    y = ptwt_amd.waverec(leaves, wavelet)
    torch.cuda.synchronize()
    rkids = [e[1] for e in _engine.level_events]
    _engine.level_events = None
    assert any(k in (15, 18) for k in rkids), rkids
    v = torch.randn_like(y)
    _engine.level_events = []
    gl = torch.autograd.grad((v * y).sum(), leaves)  # (a plain backward: the multi-level analysis launches with the rec taps reversed)
    torch.cuda.synchronize()
    bkids = [e[1] for e in _engine.level_events]
    _engine.level_events = None
    assert any(k in (14, 17) for k in bkids), bkids
    _engine.set_option(_engine.OPT_PAIR_MODE, 2)  # every multi-level launch off
    try:
        x2 = x.detach().clone().requires_grad_(True)
        c2 = ptwt_amd.wavedec(x2, wavelet, mode=mode, level=level)
        (gx2,) = torch.autograd.grad(sum((w * t).sum() for w, t in zip(ws, c2)), x2)
        leaves2 = [t.detach().clone().requires_grad_(True) for t in leaves]
        y2 = ptwt_amd.waverec(leaves2, wavelet)
        gl2 = torch.autograd.grad((v * y2).sum(), leaves2)
    finally:
        _engine.set_option(_engine.OPT_PAIR_MODE, 0)
    assert G.relerr(gx.cpu().numpy(), gx2.cpu().numpy()) < 2e-6
    assert G.relerr(y.detach().cpu().numpy(), y2.detach().cpu().numpy()) < 2e-6
    for a, b in zip(gl, gl2):
        assert G.relerr(a.cpu().numpy(), b.cpu().numpy()) < 2e-6


def test_backward_routes():
    """Zero-mode analysis adjoints and all synthesis adjoints ride on the fast kernels (kernel ids through the C ABI)."""
    import ctypes

    lib = _engine.load_library()
    d = _engine.LevelDesc()
    d.ndim, d.dtype, d.mode, d.filt_len, d.batch = 2, 0, 0, 8, 4
    for a, (n, m) in enumerate([(1024, 515), (1024, 515)]):
        d.sig_extent[a], d.coef_extent[a] = n, m
    d.sig_stride[0], d.sig_stride[1], d.sig_stride[2] = 1024 * 1024, 1024, 1
    for s in (d.approx_stride, d.detail_stride):
        s[0], s[1], s[2] = 4 * 515 * 515, 515, 1
    assert lib.mifwt_kernel_id(ctypes.byref(d), 2) == 22  # adjoint of a zero-mode analysis = a synthesis level: the streaming kernel with one level (round 4; was: id 2)
    assert lib.mifwt_kernel_id(ctypes.byref(d), 3) == 16  # adjoint of a synthesis = a zero-mode analysis level: the streaming kernel with one level (round 4; was: id 7)
    d.mode = 2
    assert lib.mifwt_kernel_id(ctypes.byref(d), 2) == 22  # reflect: the same launch + the border kernel (round 4; was: generic passes)
    assert lib.mifwt_kernel_id(ctypes.byref(d), 3) == 16
    _engine.set_option(_engine.OPT_DEBUG, 1024)
    try:
        assert lib.mifwt_kernel_id(ctypes.byref(d), 2) == 0  # the generic adjoint passes (halo fold-back), kept for short axes / long filters
    finally:
        _engine.set_option(_engine.OPT_DEBUG, 0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("mode", ["reflect", "symmetric", "periodic", "constant"])
def test_analysis_adjoint_fast_route_vs_generic_passes(mode, dtype):
    """The adjoint of an analysis level with a boundary extension (the backward of F.pad + F.conv*d, src/ptwt/conv_transform.py:135-139
    and the 2-D / 3-D twins) on its fast route — zero-mode adjoint over the whole signal + border kernel (csrc/mifwt_adjoint_border.hip)
    — against the generic per-axis adjoint passes (pinned by the reference's autograd goldens): 1-3 axes, odd / even extents, 2 .. 20
    taps, batches; and the adjoint identity <A x, g> = <x, A^T g> against the forward kernels."""
    rng = np.random.default_rng(41)
    eng = _engine.ENGINE
    tol = 2e-6 if dtype == torch.float32 else 1e-12
    for shape, wavelet in [((3, 200), "db4"), ((2, 201), "haar"), ((5, 96), "db10"), ((2, 64, 70), "db4"), ((3, 61, 128), "db2"), ((1, 300, 301), "sym8"),
                           ((2, 33, 40), "haar"), ((2, 45, 42), "db10"), ((1, 131, 67), "db3"), ((70, 18, 19), "db4"), ((1, 140, 90), "coif5"),
                           ((2, 24, 26, 31), "db2"), ((1, 40, 33, 36), "db3"), ((2, 18, 20, 19), "haar")]:
        dec_lo, dec_hi, _, _ = ptwt_amd._wavelets.host_taps(wavelet)
        flen, nd = len(dec_lo), len(shape) - 1
        x = torch.from_numpy(rng.standard_normal(shape)).to(dtype).to(dev())
        buf = eng.analysis(x, dec_lo, dec_hi, _engine.MODE_IDS[mode])
        g = torch.from_numpy(rng.standard_normal(tuple(buf.shape))).to(dtype).to(dev())
        fast = eng.analysis_adjoint(g, shape[1:], dec_lo, dec_hi, _engine.MODE_IDS[mode])
        bands = eng.analysis_adjoint_bands(g[:, 0].contiguous(), [g[:, s].contiguous() for s in range(1, 1 << nd)], shape[1:], dec_lo, dec_hi,
                                           _engine.MODE_IDS[mode])
        _engine.set_option(_engine.OPT_DEBUG, 1024)
        try:
            slow = eng.analysis_adjoint(g, shape[1:], dec_lo, dec_hi, _engine.MODE_IDS[mode])
        finally:
            _engine.set_option(_engine.OPT_DEBUG, 0)
        assert G.relerr(fast.cpu().numpy(), slow.cpu().numpy()) < tol, (shape, wavelet)
        assert float((fast - slow).abs().max()) < 50 * tol * float(slow.abs().max()), (shape, wavelet)  # (every border sample, not only the norm)
        assert G.relerr(bands.cpu().numpy(), slow.cpu().numpy()) < tol, (shape, wavelet, "per-band entry")
        if nd == 2:  # the border on the one-thread-per-sample kernel (MIFWT_OPT_DEBUG 4096) instead of one thread per border line
            _engine.set_option(_engine.OPT_DEBUG, 4096)
            try:
                per_sample = eng.analysis_adjoint(g, shape[1:], dec_lo, dec_hi, _engine.MODE_IDS[mode])
            finally:
                _engine.set_option(_engine.OPT_DEBUG, 0)
            assert float((fast - per_sample).abs().max()) < 50 * tol * float(slow.abs().max()), (shape, wavelet, "border kernels")
        lhs, rhs = (buf.double() * g.double()).sum().item(), (x.double() * fast.double()).sum().item()
        # (sums of up to 1e5 products of unit-variance numbers: |lhs| ~ 3e2, fp32 rounding of the terms ~ 1e-2 in all)
        assert abs(lhs - rhs) <= ((2e-5, 2e-2) if dtype == torch.float32 else (1e-12, 1e-9))[0] * abs(rhs) + ((2e-5, 2e-2) if dtype == torch.float32 else (1e-12, 1e-9))[1], (shape, wavelet, lhs, rhs)


def test_no_grad_and_detached_paths_unchanged():
    x = torch.randn(2, 64, 64, device=dev(), requires_grad=True)
    with torch.no_grad():
        c = ptwt_amd.wavedec2(x, "db2", level=2)
    assert not c[0].requires_grad
    c = ptwt_amd.wavedec2(x, "db2", level=2)
    assert c[0].requires_grad and c[1][0].requires_grad
    y = ptwt_amd.waverec2(c, "db2")
    y.square().sum().backward()
    # perfect reconstruction: d/dx sum(rec(dec(x))^2) = 2 x
    assert torch.allclose(x.grad, 2 * x.detach(), atol=2e-5)


def test_tap_gradients_vs_reference_autograd():
    """Gradients w.r.t. learnable filter taps (the four taps as leaf tensors, 4-tuple wavelet form) against the
    reference's own autograd (tests/golden/ptwt_ref_tapgrads.npz): analysis w.r.t. dec taps, analysis + synthesis w.r.t.
    all four — the ten level transforms, swt / iswt and the packet trees; fp64, 1e-10 norm-wise."""
    import json
    import os

    from ptwt_amd import WaveletTensorTuple

    z, idx = G.load("ptwt_ref_tapgrads.npz")
    with open(os.path.join(G.GOLDEN, "pywt_filter_banks.json")) as f:
        banks = json.load(f)
    for case in idx:
        k = case["key"]
        kw = {a: (tuple(v) if isinstance(v, list) else v) for a, v in case["kw"].items()}
        x = torch.from_numpy(z[k + "_x"]).to(dev())
        name = "haar" if case["wavelet"] == "db1" else case["wavelet"]
        taps = [torch.tensor(banks[name][f], dtype=torch.float64, device=dev(), requires_grad=True)
                for f in ("dec_lo", "dec_hi", "rec_lo", "rec_hi")]
        wt = WaveletTensorTuple(*taps)
        if case["fn"].startswith("packet"):  # packet trees: leaves of maxlevel, then reconstruct()'s root
            cls = ptwt_amd.WaveletPacket if case["fn"] == "packet1" else ptwt_amd.WaveletPacket2D
            wp = cls(x, wt, **kw)
            assert wp.get_level(kw["maxlevel"], "natural") == case["keys"]
            loss = sum((weight(wp[key], i) * wp[key]).sum() for i, key in enumerate(case["keys"]))
            g_dec = torch.autograd.grad(loss, taps[:2], retain_graph=True)
            assert G.relerr(g_dec[0].cpu().numpy(), z[k + "_gdec_lo"]) < 1e-10, (case, "dec_lo")
            assert G.relerr(g_dec[1].cpu().numpy(), z[k + "_gdec_hi"]) < 1e-10, (case, "dec_hi")
            wp.reconstruct()
            y = wp[""]
            g_all = torch.autograd.grad((weight(y, 7) * y).sum(), taps)
            for nme, g in zip(("dec_lo", "dec_hi", "rec_lo", "rec_hi"), g_all):
                assert G.relerr(g.cpu().numpy(), z["%s_gall_%s" % (k, nme)]) < 1e-10, (case, nme)
            continue
        coeffs = getattr(ptwt_amd, case["fn"])(x, wt, **kw)
        fl = flat(coeffs)
        loss = sum((weight(t, i) * t).sum() for i, t in enumerate(fl))
        g_dec = torch.autograd.grad(loss, taps[:2], retain_graph=True)
        assert G.relerr(g_dec[0].cpu().numpy(), z[k + "_gdec_lo"]) < 1e-10, (case, "dec_lo")
        assert G.relerr(g_dec[1].cpu().numpy(), z[k + "_gdec_hi"]) < 1e-10, (case, "dec_hi")
        rkw = {a: v for a, v in kw.items() if a in ("axis", "axes")}
        y = getattr(ptwt_amd, case["rec"])(coeffs, wt, **rkw)
        g_all = torch.autograd.grad((weight(y, 7) * y).sum(), taps)
        for nme, g in zip(("dec_lo", "dec_hi", "rec_lo", "rec_hi"), g_all):
            assert g.shape == taps[0].shape
            assert G.relerr(g.cpu().numpy(), z["%s_gall_%s" % (k, nme)]) < 1e-10, (case, nme)


def test_learnable_wavelet_module_trains():
    """A pywt-like object whose filter bank are nn.Parameters (as the reference's learnable wavelets expose them,
    src/ptwt/wavelets_learnable.py:167-277): one SGD step on the perfect-reconstruction loss moves all four filters."""
    class Learnable(torch.nn.Module):
        def __init__(self):
            super().__init__()
            import json, os
            with open(os.path.join(G.GOLDEN, "pywt_filter_banks.json")) as f:
                bank = json.load(f)["db2"]
            self.taps = torch.nn.ParameterList([torch.nn.Parameter(torch.tensor(bank[n], dtype=torch.float32) + 0.01 * i)
                                                for i, n in enumerate(("dec_lo", "dec_hi", "rec_lo", "rec_hi"))])

        @property
        def filter_bank(self):
            return tuple(self.taps)

        def __len__(self):
            return self.taps[0].shape[0]

    w = Learnable().to(dev())
    opt = torch.optim.SGD(w.parameters(), lr=1e-2)
    x = torch.randn(4, 64, 64, device=dev())
    before = [p.detach().clone() for p in w.parameters()]
    rec = ptwt_amd.waverec2(ptwt_amd.wavedec2(x, w, level=2, mode="periodic"), w)
    loss = (rec[..., :64, :64] - x).square().mean()
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() and p.grad.abs().max() > 0 for p in w.parameters())
    opt.step()
    assert all(not torch.equal(b, p.detach()) for b, p in zip(before, w.parameters()))


@pytest.mark.parametrize("fn,rec,shape,mode", [("wavedec", "waverec", (3, 200), "reflect"), ("wavedec2", "waverec2", (2, 40, 52), "symmetric"),
                                                ("wavedec3", "waverec3", (1, 20, 18, 22), "zero"), ("fswavedec2", "fswaverec2", (2, 36, 40), "periodic")])
def test_second_order_gradients(fn, rec, shape, mode):
    """Gradients of gradients w.r.t. the data (gradient penalties, Hessian-vector products): the reference has them because it
    is plain F.pad + conv / conv_transpose; here every adjoint is a differentiable level op in turn.  The transform is linear
    (c = A x), so for f(x) = sum w_i c_i^2 / 2 the Hessian-vector product is H v = A^T diag(w) A v = the FIRST-order gradient of
    f at v: both sides come from the same kernels, only the route through autograd differs."""
    torch.manual_seed(3)
    x = torch.randn(*shape, device=dev(), dtype=torch.float64, requires_grad=True)
    v = torch.randn(*shape, device=dev(), dtype=torch.float64)

    def f(t):
        cs = flat(getattr(ptwt_amd, fn)(t, "db3", level=2, mode=mode))
        return sum((weight(c, i) * c.square()).sum() for i, c in enumerate(cs)) / 2

    (g,) = torch.autograd.grad(f(x), x, create_graph=True)
    (hv,) = torch.autograd.grad((g * v).sum(), x)
    vv = v.clone().requires_grad_(True)
    (want,) = torch.autograd.grad(f(vv), vv)
    assert G.relerr(hv.cpu().numpy(), want.cpu().numpy()) < 1e-11
    # and through the synthesis side: y = R c, d/dc of <grad_c(sum y^2 / 2), u> = R^T R u
    cs = [c.detach().clone().requires_grad_(True) for c in flat(getattr(ptwt_amd, fn)(x.detach(), "db3", level=2, mode=mode))]
    coeffs = rebuild(getattr(ptwt_amd, fn)(x.detach(), "db3", level=2, mode=mode), cs)
    y = getattr(ptwt_amd, rec)(coeffs, "db3")
    gs = torch.autograd.grad(y.square().sum() / 2, cs, create_graph=True)
    us = [torch.randn_like(c) for c in cs]
    hu = torch.autograd.grad(sum((a * b).sum() for a, b in zip(gs, us)), cs)
    cu = [u.clone().requires_grad_(True) for u in us]
    yu = getattr(ptwt_amd, rec)(rebuild(coeffs, cu), "db3")
    want = torch.autograd.grad(yu.square().sum() / 2, cu)
    for a, b in zip(hu, want):
        assert G.relerr(a.cpu().numpy(), b.cpu().numpy()) < 1e-11


def test_second_order_gradients_swt():
    torch.manual_seed(4)
    x = torch.randn(3, 96, device=dev(), dtype=torch.float64, requires_grad=True)
    v = torch.randn(3, 96, device=dev(), dtype=torch.float64)

    def f(t):
        return sum((weight(c, i) * c.square()).sum() for i, c in enumerate(ptwt_amd.swt(t, "db2", 3))) / 2

    (g,) = torch.autograd.grad(f(x), x, create_graph=True)
    (hv,) = torch.autograd.grad((g * v).sum(), x)
    vv = v.clone().requires_grad_(True)
    (want,) = torch.autograd.grad(f(vv), vv)
    assert G.relerr(hv.cpu().numpy(), want.cpu().numpy()) < 1e-11
    cs = [c.detach().clone().requires_grad_(True) for c in ptwt_amd.swt(x.detach(), "db2", 3)]
    gs = torch.autograd.grad(ptwt_amd.iswt(cs, "db2").square().sum() / 2, cs, create_graph=True)
    us = [torch.randn_like(c) for c in cs]
    hu = torch.autograd.grad(sum((a * b).sum() for a, b in zip(gs, us)), cs)
    cu = [u.clone().requires_grad_(True) for u in us]
    want = torch.autograd.grad(ptwt_amd.iswt(cu, "db2").square().sum() / 2, cu)
    for a, b in zip(hu, want):
        assert G.relerr(a.cpu().numpy(), b.cpu().numpy()) < 1e-11


def weight2(t, i):
    return torch.cos(0.53 * torch.arange(t.numel(), dtype=torch.float64, device=t.device) + i).reshape(t.shape).to(t.dtype)


def test_second_order_gradients_with_learnable_taps_vs_reference():
    """create_graph=True through a learnable filter bank (round 4): the mixed second derivatives — data x taps, taps x
    taps, upstream gradient x taps — against the reference's own double backward through ATen's conv path
    (tests/golden/ptwt_ref_tapgrads2.npz, tests/golden/make_ptwt_ref_tapgrad2_goldens.py): the ten level transforms, 1-3 axes, five
    boundary modes, and swt / iswt; fp64, 1e-9 norm-wise.  (The backward that is asked for a graph re-runs the level as a product of per-axis ops that
    are closed under differentiation, `_fwt._Axis1` / `_Syn1`.)"""
    import json
    import os

    from ptwt_amd import WaveletTensorTuple

    z, idx = G.load("ptwt_ref_tapgrads2.npz")
    with open(os.path.join(G.GOLDEN, "pywt_filter_banks.json")) as f:
        banks = json.load(f)
    for case in idx:
        k = case["key"]
        kw = {a: (tuple(v) if isinstance(v, list) else v) for a, v in case["kw"].items()}
        x = torch.from_numpy(z[k + "_x"]).to(dev()).requires_grad_(True)
        name = "haar" if case["wavelet"] == "db1" else case["wavelet"]
        taps = [torch.tensor(banks[name][f], dtype=torch.float64, device=dev(), requires_grad=True) for f in ("dec_lo", "dec_hi", "rec_lo", "rec_hi")]
        wt = WaveletTensorTuple(*taps)
        # ---- analysis
        fl = flat(getattr(ptwt_amd, case["fn"])(x, wt, **kw))
        f = sum((weight(t, i) * t.square()).sum() for i, t in enumerate(fl)) / 2
        g_x, t_lo, t_hi = torch.autograd.grad(f, [x, taps[0], taps[1]], create_graph=True)
        s1 = (g_x * weight2(g_x, 1)).sum() + (t_lo * weight2(t_lo, 2)).sum() + (t_hi * weight2(t_hi, 3)).sum()
        d = torch.autograd.grad(s1, [x, taps[0], taps[1]])
        for got, nme in zip(d, ("a_dx", "a_dlo", "a_dhi")):
            assert G.relerr(got.cpu().numpy(), z["%s_%s" % (k, nme)]) < 1e-9, (case, nme)
        # ---- synthesis
        coeffs = getattr(ptwt_amd, case["fn"])(x.detach(), case["wavelet"], **kw)
        leaves = [t.detach().clone().requires_grad_(True) for t in flat(coeffs)]
        assert len(leaves) == case["ncoef"]
        rkw = {a: v for a, v in kw.items() if a in ("axis", "axes")}
        y = getattr(ptwt_amd, case["rec"])(rebuild(coeffs, leaves), wt, **rkw)
        f = (weight(y, 7) * y.square()).sum() / 2
        grads = torch.autograd.grad(f, leaves + [taps[2], taps[3]], create_graph=True)
        s2 = sum((gc * weight2(gc, 4 + i)).sum() for i, gc in enumerate(grads[:-2]))
        s2 = s2 + (grads[-2] * weight2(grads[-2], 2)).sum() + (grads[-1] * weight2(grads[-1], 3)).sum()
        d2 = torch.autograd.grad(s2, leaves + [taps[2], taps[3]])
        for i, got in enumerate(d2[:-2]):
            assert G.relerr(got.cpu().numpy(), z["%s_s_dc%d" % (k, i)]) < 1e-9, (case, "s_dc", i)
        assert G.relerr(d2[-2].cpu().numpy(), z[k + "_s_dlo"]) < 1e-9, (case, "s_dlo")
        assert G.relerr(d2[-1].cpu().numpy(), z[k + "_s_dhi"]) < 1e-9, (case, "s_dhi")


def test_double_backward_through_learnable_taps_works_third_order_refused():
    """A gradient penalty with a learnable wavelet (create_graph=True) works for the decimated transforms, the packet trees and the
    stationary transform (round 3 refused all of them); a THIRD-order derivative through learnable taps raises instead of returning a
    graph that lacks terms."""
    bank = tuple(torch.tensor(v, device=dev(), dtype=torch.float64, requires_grad=True) for v in ptwt_amd._wavelets.host_taps("db2"))
    x = torch.randn(2, 32, 32, device=dev(), dtype=torch.float64, requires_grad=True)
    y = sum(c.square().sum() if isinstance(c, torch.Tensor) else sum(t.square().sum() for t in c) for c in ptwt_amd.wavedec2(x, bank, level=1))
    (g,) = torch.autograd.grad(y, x, create_graph=True)
    pen = g.square().sum()
    gb = torch.autograd.grad(pen, [x, bank[0], bank[1]], retain_graph=True)
    assert all(torch.isfinite(t).all() and t.abs().sum() > 0 for t in gb)
    with pytest.raises(RuntimeError, match="beyond second order"):  # a graph of the second derivatives = third order
        torch.autograd.grad(pen, [x, bank[0], bank[1]], create_graph=True)
    wp = ptwt_amd.WaveletPacket(x[:, 0], bank, mode="reflect", maxlevel=2)
    leaf = wp["ad"]
    (g,) = torch.autograd.grad(leaf.square().sum(), x, create_graph=True)
    gb = torch.autograd.grad(g.square().sum(), [bank[0], bank[1]])
    assert all(torch.isfinite(t).all() and t.abs().sum() > 0 for t in gb)
    xs = torch.randn(2, 64, device=dev(), dtype=torch.float64, requires_grad=True)
    ys = sum(c.square().sum() for c in ptwt_amd.swt(xs, bank, level=2))
    (g,) = torch.autograd.grad(ys, xs, create_graph=True)
    gb = torch.autograd.grad(g.square().sum(), [xs, bank[0], bank[1]])
    assert all(torch.isfinite(t).all() and t.abs().sum() > 0 for t in gb)


def test_tensor_taps_are_read_live_on_every_call():
    """Tap tensors on the GPU are read back on every top-level call, like the reference reads the live tensor: an update through
    ``.data`` (which does not bump the tensor's version counter) must change the next transform."""
    from ptwt_amd import _wavelets

    t = torch.tensor([0.5, 0.5], device=dev())
    assert _wavelets._to_floats(t) == (0.5, 0.5)
    t.data.mul_(2.0)
    assert _wavelets._to_floats(t) == (1.0, 1.0)
    p = tuple(torch.nn.Parameter(torch.tensor(v, device=dev(), dtype=torch.float64)) for v in ptwt_amd._wavelets.host_taps("db2"))
    xx = torch.randn(2, 64, device=dev(), dtype=torch.float64)
    with torch.no_grad():
        before = ptwt_amd.wavedec(xx, p, level=2)
        for q in p:
            q.data.mul_(2.0)
        after = ptwt_amd.wavedec(xx, p, level=2)
    assert torch.allclose(after[0], 4.0 * before[0], rtol=1e-12, atol=0)  # two levels, every filter doubled
    assert torch.allclose(after[-1], 2.0 * before[-1], rtol=1e-12, atol=0)
    bank = tuple(torch.tensor(v, device=dev(), dtype=torch.float64) for v in ptwt_amd._wavelets.host_taps("db2"))
    x = torch.randn(2, 64, device=dev(), dtype=torch.float64)
    c1 = ptwt_amd.wavedec(x, bank, level=2)
    c2 = ptwt_amd.wavedec(x, "db2", level=2)
    for u, w in zip(c1, c2):
        assert torch.equal(u, w)


def test_learnable_taps_stay_on_the_gpu_no_sync_and_capturable():
    """Device-resident taps (round 5).  A filter bank of leaf tensors on the GPU — the reference's learnable
    wavelets keep their taps as tensors in the graph, src/ptwt/_util.py:115-132; examples/network_compression/wavelet_linear.py:118,150
    — reaches the kernels as device memory (C ABI mifwt_*_dtaps): forward AND backward of the ten level transforms run without a
    single host synchronisation (`torch.cuda.set_sync_debug_mode("error")` turns any into an exception), and the gradients still
    match the reference's own autograd (tests/golden/ptwt_ref_tapgrads.npz; fp64, 1e-10 norm-wise).  A forward with device taps can
    also be captured into a HIP graph; after an in-place update of the taps the replay sees the new filter."""
    import json
    import os

    from ptwt_amd import WaveletTensorTuple

    z, idx = G.load("ptwt_ref_tapgrads.npz")
    with open(os.path.join(G.GOLDEN, "pywt_filter_banks.json")) as f:
        banks = json.load(f)
    ran = 0
    for case in idx:
        if case["fn"].startswith("packet") or case["fn"] in ("swt", "iswt"):
            continue  # (packet trees and the stationary transform read their taps on the host: other kernels)
        k = case["key"]
        kw = {a: (tuple(v) if isinstance(v, list) else v) for a, v in case["kw"].items()}
        x = torch.from_numpy(z[k + "_x"]).to(dev())
        name = "haar" if case["wavelet"] == "db1" else case["wavelet"]
        taps = [torch.tensor(banks[name][f], dtype=torch.float64, device=dev(), requires_grad=True)
                for f in ("dec_lo", "dec_hi", "rec_lo", "rec_hi")]
        wt = WaveletTensorTuple(*taps)
        rkw = {a: v for a, v in kw.items() if a in ("axis", "axes")}
        # weights of the losses, made outside the no-sync region (shapes from a first, ordinary call)
        with torch.no_grad():
            c0 = getattr(ptwt_amd, case["fn"])(x, wt, **kw)
            w_c = [weight(t, i) for i, t in enumerate(flat(c0))]
            w_y = weight(getattr(ptwt_amd, case["rec"])(c0, wt, **rkw), 7)
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode("error")
        try:
            coeffs = getattr(ptwt_amd, case["fn"])(x, wt, **kw)
            loss = sum((w * t).sum() for w, t in zip(w_c, flat(coeffs)))
            g_dec = torch.autograd.grad(loss, taps[:2], retain_graph=True)
            y = getattr(ptwt_amd, case["rec"])(coeffs, wt, **rkw)
            g_all = torch.autograd.grad((w_y * y).sum(), taps)
        finally:
            torch.cuda.set_sync_debug_mode("default")
        assert G.relerr(g_dec[0].cpu().numpy(), z[k + "_gdec_lo"]) < 1e-10, (case, "dec_lo")
        assert G.relerr(g_dec[1].cpu().numpy(), z[k + "_gdec_hi"]) < 1e-10, (case, "dec_hi")
        for nme, g in zip(("dec_lo", "dec_hi", "rec_lo", "rec_hi"), g_all):
            assert G.relerr(g.cpu().numpy(), z["%s_gall_%s" % (k, nme)]) < 1e-10, (case, nme)
        ran += 1
    assert ran >= 20, ran

    # negative control: with device taps switched off the same call reads the bank back, and the sync detector says so
    xs = torch.randn(2, 40, 52, dtype=torch.float64).to(dev())
    lt = [torch.tensor(banks["db2"][f], dtype=torch.float64, device=dev(), requires_grad=True) for f in ("dec_lo", "dec_hi", "rec_lo", "rec_hi")]
    ptwt_amd.set_device_taps("never")
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        with pytest.raises(RuntimeError):
            ptwt_amd.wavedec2(xs, tuple(lt), level=1)
    finally:
        torch.cuda.set_sync_debug_mode("default")
        ptwt_amd.set_device_taps("auto")

    # values: device taps against host taps (fused kernels), forward only; and capture + replay after an in-place tap update
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 96, 130, generator=g).to(dev())
    bank = [torch.tensor(banks["db3"][f], dtype=torch.float32, device=dev()) for f in ("dec_lo", "dec_hi", "rec_lo", "rec_hi")]
    ptwt_amd.set_device_taps("always")
    try:
        got = ptwt_amd.wavedec2(x, tuple(bank), level=2, mode="symmetric")
        cap = ptwt_amd.capture(lambda t: ptwt_amd.waverec2(ptwt_amd.wavedec2(t, tuple(bank), level=2, mode="symmetric"), tuple(bank)), x)
        y0 = [cap(x).clone()]
        with torch.no_grad():
            bank[0].mul_(1.25)
        y1 = [cap(x).clone()]
        want1 = ptwt_amd.waverec2(ptwt_amd.wavedec2(x, tuple(bank), level=2, mode="symmetric"), tuple(bank))
        with torch.no_grad():
            bank[0].div_(1.25)
    finally:
        ptwt_amd.set_device_taps("auto")
    want = ptwt_amd.wavedec2(x, "db3", level=2, mode="symmetric")
    for (n, a), (_, b) in zip(G.flatten_coeffs(got), G.flatten_coeffs(want)):
        assert G.relerr(a.cpu().numpy(), b.cpu().double().numpy()) < 2e-6, n
    assert not torch.equal(y0[0], y1[0]), "the replay did not see the updated taps"
    assert torch.equal(y1[0], want1), "replay after the update differs from an eager call with the updated taps"


def test_learnable_taps_run_on_the_fused_kernels_bit_identical_to_host_taps():
    """Round 6: a learnable filter bank on the GPU no longer leaves the fused kernels.  The per-level ops of a
    differentiable call with tap tensors reach the LDS-tile kernels (ids 7 / 8), one level of the streaming kernels (ids 16 / 22) and the
    border kernels with the taps as DEVICE memory (C ABI mifwt_*_dtaps, DevTapArg): same kernel ids as the host-tap call
    (``level_events``), results and data gradients BIT-identical to it (a kernel reads the doubles once and converts them as the host
    does), tap gradients equal to rounding of the correlation kernel's atomic sums (it sees the same inputs), no host synchronisation."""
    import json
    import os

    from ptwt_amd import _engine

    with open(os.path.join(G.GOLDEN, "pywt_filter_banks.json")) as f:
        banks = json.load(f)
    g = torch.Generator().manual_seed(11)
    cases = [  # (shape, wavelet, mode, level, dtype, kernel ids the forward / backward must show)
        ((2, 1024, 1024), "db4", "reflect", 2, torch.float32, {16, 7}, {22, 8}),
        ((3, 300, 260), "db3", "symmetric", 2, torch.float32, {7}, {8}),
        ((2, 257, 300), "db2", "zero", 2, torch.float64, {7}, {8}),
        ((2, 200, 180), "sym5", "constant", 1, torch.float32, {7}, {8}),
    ]
    for shape, wav, mode, level, dtype, kf, kb in cases:
        x = torch.randn(*shape, generator=g, dtype=dtype).to(dev())
        res = {}
        for how in ("device", "host"):
            ptwt_amd.set_device_taps("auto" if how == "device" else "never")
            try:
                taps = [torch.tensor(banks[wav][f], dtype=torch.float64, device=dev(), requires_grad=True) for f in ("dec_lo", "dec_hi", "rec_lo", "rec_hi")]
                xx = x.clone().requires_grad_(True)
                with torch.no_grad():
                    c0 = ptwt_amd.wavedec2(xx, tuple(taps), mode=mode, level=level)
                    w_c = [weight(t, i) for i, t in enumerate(flat(c0))]
                    w_y = weight(ptwt_amd.waverec2(c0, tuple(taps)), 3)
                torch.cuda.synchronize()
                _engine.level_events = []
                if how == "device":
                    torch.cuda.set_sync_debug_mode("error")
                try:
                    coeffs = ptwt_amd.wavedec2(xx, tuple(taps), mode=mode, level=level)
                    y = ptwt_amd.waverec2(coeffs, tuple(taps))
                    loss = sum((w * t).sum() for w, t in zip(w_c, flat(coeffs))) + (w_y * y).sum()
                    grads = torch.autograd.grad(loss, [xx] + taps)
                finally:
                    torch.cuda.set_sync_debug_mode("default")
                ev = _engine.level_events
                _engine.level_events = None
                res[how] = ([t.detach().clone() for t in flat(coeffs)] + [y.detach().clone()], [t.clone() for t in grads],
                            {e[1] for e in ev if e[0] == "fwd"}, {e[1] for e in ev if e[0] == "inv"}, {e[1] for e in ev if e[0].endswith("_adj")})
            finally:
                ptwt_amd.set_device_taps("auto")
        (vd, gd, fd, idv, ad), (vh, gh, fh, ih, ah) = res["device"], res["host"]
        # the same kernels as the host-tap call, the fused ones among them, NOTHING on the generic passes (id 0) — forward levels, the
        # adjoints, and the 1-D per-axis maps of the tap gradients (streaming axis kernels, ids 3 / 4)
        assert fd == fh and idv == ih and ad == ah, (shape, wav, fd, fh, idv, ih, ad, ah)
        assert kf <= fd and kb <= idv and 0 not in (fd | idv | ad), (shape, wav, fd, idv, ad)
        for a, b in zip(vd, vh):
            assert torch.equal(a, b), (shape, wav, mode, "values")
        assert torch.equal(gd[0], gh[0]), (shape, wav, mode, "data gradient")
        for a, b in zip(gd[1:], gh[1:]):  # (the correlation kernel sums with atomics: the same inputs, not the same order of additions)
            assert G.relerr(a.cpu().numpy(), b.cpu().numpy()) < 1e-12, (shape, wav, mode, "tap gradient")


def test_natural_layout_tap_gradient_kernels_vs_transposed_row_kernels():
    """Round 6: the tap gradients of a 2-D level run on operands in their natural layout — ``tap_correlate_planes`` (along the columns:
    the row kernel with a two-level row index; along the rows: the column kernel with its sliding register window), the outer-axis
    level kernels on their own (``analysis_outer`` / ``synthesis_outer``).  Each against what the host layer did before: transposed
    copies in front of the row kernels (whose results the reference's tap-gradient goldens pin).  Every mode, both signs, odd extents,
    strided views (planes of a level buffer), f32 / f64, 2 .. 20 taps."""
    from ptwt_amd import _engine

    eng = _engine.ENGINE
    g = torch.Generator().manual_seed(23)
    for dtype, tol in ((torch.float64, 1e-12), (torch.float32, 2e-5)):
        for flen in (2, 4, 8, 10, 16, 20):
            for mode in ("zero", "constant", "reflect", "periodic", "symmetric"):
                mid = _engine.MODE_IDS[mode]
                B, H, W = 3, 2 * flen + 9, 2 * flen + 70
                Mh, Mw = (H + flen - 1) // 2, (W + flen - 1) // 2
                big = torch.randn(B, 4, H, W, generator=g, dtype=dtype).to(dev())
                x = big[:, 1]  # a strided view: batch stride 4 H W
                gb = torch.randn(B, 4, Mh, Mw, generator=g, dtype=dtype).to(dev())
                lo = [float(v) for v in torch.randn(flen, generator=g, dtype=torch.float64)]
                hi = [float(v) for v in torch.randn(flen, generator=g, dtype=torch.float64)]
                # the outer-axis level against the inner-axis level of the transposed planes
                zl, zh = eng.analysis_outer(x, lo, hi, mid)
                xt = x.transpose(1, 2).contiguous()
                ref = eng.analysis(xt.reshape(B * W, H), lo, hi, mid).reshape(B, W, 2, Mh)
                assert torch.equal(zl, ref[:, :, 0].transpose(1, 2)) and torch.equal(zh, ref[:, :, 1].transpose(1, 2)), (dtype, flen, mode)
                for sgn, c0, m_id in ((-1, 1, mid), (1, -(flen - 2), 0)):
                    # along the columns: a [B, R, M], b [B, R, N]
                    a, b = gb[:, 2], (zl if sgn < 0 else torch.randn(B, Mh, 2 * Mw - flen + 2, generator=g, dtype=dtype).to(dev()))
                    got = torch.zeros(flen, dtype=torch.float64, device=dev())
                    want = torch.zeros_like(got)
                    eng.tap_correlate_planes(1, a, b, flen, c0, sgn, m_id, got)
                    eng.tap_correlate(a.reshape(-1, a.shape[-1]), b.reshape(-1, b.shape[-1]), flen, c0, sgn, m_id, want)
                    assert G.relerr(got.cpu().numpy(), want.cpu().numpy()) < tol, (dtype, flen, mode, sgn, "columns")
                    # along the rows: a [B, M, C], b [B, N, C]
                    a2 = gb[:, 1]
                    b2 = torch.randn(B, H if sgn < 0 else 2 * Mh - flen + 2, Mw, generator=g, dtype=dtype).to(dev())
                    got = torch.zeros(flen, dtype=torch.float64, device=dev())
                    want = torch.zeros_like(got)
                    eng.tap_correlate_planes(0, a2, b2, flen, c0, sgn, m_id, got)
                    at, bt = a2.transpose(1, 2).contiguous(), b2.transpose(1, 2).contiguous()
                    eng.tap_correlate(at.reshape(-1, at.shape[-1]), bt.reshape(-1, bt.shape[-1]), flen, c0, sgn, m_id, want)
                    assert G.relerr(got.cpu().numpy(), want.cpu().numpy()) < tol, (dtype, flen, mode, sgn, "rows")
                # the outer-axis synthesis against the inner-axis synthesis of the transposed planes (both crops)
                for n_out in (2 * Mh - flen + 2, 2 * Mh - flen + 1):
                    if n_out < 1:
                        continue
                    y = eng.synthesis_outer(gb[:, 0], gb[:, 3], lo, hi, n_out)
                    lt, ht = gb[:, 0].transpose(1, 2).contiguous(), gb[:, 3].transpose(1, 2).contiguous()
                    yr = eng.synthesis(lt.reshape(B * Mw, Mh), [ht.reshape(B * Mw, Mh)], lo, hi, [n_out]).reshape(B, Mw, n_out).transpose(1, 2)
                    assert torch.equal(y, yr), (dtype, flen, mode, n_out)
