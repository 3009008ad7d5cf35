"""Host side of the padded-convolution FWT: one N-D implementation behind the ten ptwt entry points.

What the reference spreads over conv_transform{,_2,_3}.py and separable_conv_transform.py is a single
level loop here, parameterised by the number of transformed axes; every level is one call into
``libmifwt.so`` (:mod:`._engine`).  This module restates only the thin glue: argument checks and their
error types, moving the transformed axes last and folding the batch, default levels, the per-level
extent / trim arithmetic, and the return containers — so results, shapes, containers and errors match the
reference (citations inline; /root/reference = ptwt 1.0.2-dev).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple, Union

import torch

from . import _engine
from . import _wavelets
from ._wavelets import device_bank, dwtn_max_level, filter_length, host_taps
from .constants import WaveletDetailTuple2d, supported_dtypes

AxisHint = Union[int, Sequence[int], None]


# ------------------------------------------------------------------------------------------ autograd
def _rows_last(t: torch.Tensor, axis: int) -> torch.Tensor:
    """[B, e_0.., e_{n-1}] -> 2-D [rows, e_axis] with ``axis`` (0-based among the trailing dims) innermost."""
    t = t.movedim(1 + axis, -1)
    return t.reshape(-1, t.shape[-1])


def _band_index(ndim: int, axis: int, sigma: int, rest: int) -> int:
    """Band index of the level buffer: bit (ndim-1-a) <=> axis a high-pass; ``rest`` enumerates the other axes' bits in
    axis order (most significant first)."""
    s, r = 0, rest
    for a in reversed(range(ndim)):
        if a == axis:
            bit = sigma
        else:
            bit = r & 1
            r >>= 1
        s |= bit << (ndim - 1 - a)
    return s


def _analysis_tap_grads(x, g_buf, dec_lo, dec_hi, mode_id):
    """d loss / d (dec_lo, dec_hi) of one analysis level (float64 [L] each).  Along axis a the level is
    c[k] = sum_m h[m] z_ext[2k+1-m] with z = the input transformed over the OTHER axes (ordinary level calls with axis a
    folded into the batch); the reduction is ``mifwt_tap_correlate`` (reference: ATen conv backward w.r.t. the weight)."""
    nd = x.dim() - 1
    flen = len(dec_lo)
    g_lo = torch.zeros(flen, dtype=torch.float64, device=x.device)
    g_hi = torch.zeros_like(g_lo)
    eng = _engine.ENGINE
    if nd == 2 and flen <= 32 and hasattr(eng, "tap_correlate_planes"):
        # Two axes, every operand in its NATURAL layout (round 6: the transposed copies in front of the row reductions were a third of
        # a training step).  Along the rows axis H: z = the level along W of every row (inner-axis kernel), reduced along H by the
        # column kernel; along W: z = the level along H (outer-axis kernel on its own), reduced along W by the row kernel.  The band
        # planes of g_buf are strided views, taken as they are.
        B, H, W = x.shape
        xc = x if x.stride(-1) == 1 else x.contiguous()
        rows = xc.reshape(B * H, W)  # (a view of a dense input; a copy of a plane of a level buffer: the deeper levels, a quarter each)
        pb = eng.analysis(rows, dec_lo, dec_hi, mode_id).reshape(B, H, 2, -1)  # [B, H, (lo, hi along W), Mw]
        for r in range(2):
            z = pb[:, :, r]  # [B, H, Mw]
            for sigma, out in ((0, g_lo), (1, g_hi)):
                eng.tap_correlate_planes(0, g_buf[:, _band_index(2, 0, sigma, r)], z, flen, 1, -1, mode_id, out)
        for r, z in enumerate(eng.analysis_outer(xc, dec_lo, dec_hi, mode_id)):  # (lo, hi along H) [B, Mh, W]
            for sigma, out in ((0, g_lo), (1, g_hi)):
                eng.tap_correlate_planes(1, g_buf[:, _band_index(2, 1, sigma, r)], z, flen, 1, -1, mode_id, out)
        return g_lo, g_hi
    for a in range(nd):
        if nd == 1:
            parts = [x]  # [B, N]
            part_axis = 0
        else:
            xa = x.movedim(1 + a, 1)  # [B, N_a, others..]
            flat = xa.reshape(-1, *xa.shape[2:])
            pb = eng.analysis(flat, dec_lo, dec_hi, mode_id)  # [B * N_a, 2^(nd-1), M_others..]
            pb = pb.reshape(xa.shape[0], xa.shape[1], *pb.shape[1:])  # [B, N_a, 2^(nd-1), M_others..]
            parts = [pb[:, :, r] for r in range(1 << (nd - 1))]  # each [B, N_a, M_others..]
            part_axis = 0
        for r, z in enumerate(parts):
            z2 = _rows_last(z, part_axis) if nd > 1 else z
            for sigma, out in ((0, g_lo), (1, g_hi)):
                band = _band_index(nd, a, sigma, r)
                g2 = _rows_last(g_buf[:, band], a)
                eng.tap_correlate(g2, z2, flen, 1, -1, mode_id, out)
    return g_lo, g_hi


def _synthesis_tap_grads(g_y, approx, details, rec_lo, rec_hi):
    """d loss / d (rec_lo, rec_hi) of one synthesis level.  Along axis a: y[n] = sum_k u[k] g[n+L-2-2k] with u = the
    bands whose axis-a letter is lo / hi, synthesised over the OTHER axes."""
    nd = approx.dim() - 1
    flen = len(rec_lo)
    g_lo = torch.zeros(flen, dtype=torch.float64, device=g_y.device)
    g_hi = torch.zeros_like(g_lo)
    eng = _engine.ENGINE
    bands = [approx] + list(details)
    if nd == 2 and flen <= 32 and hasattr(eng, "synthesis_outer"):
        # two axes, every operand in its natural layout (see _analysis_tap_grads).  Along H: u = the two bands with letter sigma along H
        # synthesised along W (inner-axis kernel, rows (b, k_h)), reduced along H by the column kernel; along W: u = the two bands with
        # letter sigma along W synthesised along H (outer-axis kernel), reduced along W by the row kernel.
        B, Ny, Nx = g_y.shape
        gyc = g_y if g_y.stride(-1) == 1 else g_y.contiguous()
        for sigma, out in ((0, g_lo), (1, g_hi)):
            lo_w, hi_w = bands[_band_index(2, 0, sigma, 0)], bands[_band_index(2, 0, sigma, 1)]  # [B, Mh, Mw]: low / high along W
            Mh, Mw = lo_w.shape[1:]
            u = eng.synthesis(lo_w.reshape(B * Mh, Mw), [hi_w.reshape(B * Mh, Mw)], rec_lo, rec_hi, [Nx]).reshape(B, Mh, Nx)
            eng.tap_correlate_planes(0, u, gyc, flen, -(flen - 2), 1, 0, out)
            lo_h, hi_h = bands[_band_index(2, 1, sigma, 0)], bands[_band_index(2, 1, sigma, 1)]  # low / high along H
            u = eng.synthesis_outer(lo_h, hi_h, rec_lo, rec_hi, Ny)  # [B, Ny, Mw]
            eng.tap_correlate_planes(1, u, gyc, flen, -(flen - 2), 1, 0, out)
        return g_lo, g_hi
    for a in range(nd):
        gy2 = _rows_last(g_y, a)
        for sigma, out in ((0, g_lo), (1, g_hi)):
            if nd == 1:
                u2 = bands[sigma]
            else:
                sel = [bands[_band_index(nd, a, sigma, r)].movedim(1 + a, 1) for r in range(1 << (nd - 1))]  # [B, M_a, M_others..]
                flat = [t.reshape(-1, *t.shape[2:]).contiguous() for t in sel]
                out_ext = [g_y.shape[1 + o] for o in range(nd) if o != a]
                u = eng.synthesis(flat[0], flat[1:], rec_lo, rec_hi, out_ext)  # [B * M_a, N_others..]
                u = u.reshape(sel[0].shape[0], sel[0].shape[1], *u.shape[1:])  # [B, M_a, N_others..]
                u2 = _rows_last(u, 0)
            eng.tap_correlate(u2, gy2, flen, -(flen - 2), 1, 0, out)
    return g_lo, g_hi


# ---- learnable filter banks, gradients of any order -----------------------------------------------------------------------------------
# A level along ONE axis is bilinear in (signal, taps):  c = A(h) x.  Three maps close that under differentiation, each a differentiable
# op whose backward is made of the other two:
#     A(h) x            (_Axis1)      d/dx: A(h)^T g              d/dh: C(x, g)
#     A(h)^T g          (_Axis1Adj)   d/dg: A(h) gg               d/dh: C(gg, g)
#     C(x, g) = t       (_Axis1Corr)  d/dx: A(w)^T g              d/dg: A(w) x          (w = the gradient that arrives for t: taps!)
# and likewise for the synthesis  y = S(r) (a, d)  (_Syn1, _Syn1Adj, _Syn1Corr).  An N-D level is a product of such maps, one per axis,
# so autograd through their composition yields every mixed derivative the reference gets from plain ATen ops (src/ptwt/_util.py:115-132
# keeps the taps in the graph).  The fused level kernels serve the forward and the first-order backward as before; only a backward
# that is asked for a graph (create_graph=True) with a learnable filter bank re-runs the level through these ops.  Taps travel to the
# kernels as host floats (one device-to-host copy per op).
def _host_floats_of(t: torch.Tensor) -> List[float]:
    """Host copy of a filter tensor (the stationary-transform kernels take their taps in the launch arguments)."""
    return [float(v) for v in t.detach().double().cpu().reshape(-1).tolist()]


def _host_taps_of(t: torch.Tensor, device: Optional[torch.device] = None):
    """The taps of a filter tensor for a level call on data that lives on ``device``: the tensor itself (kernels read device memory) when
    it lives on THAT GPU and device taps are not switched off, else host floats (one device-to-host copy) — a kernel launched on cuda:1
    must not be handed a raw cuda:0 pointer."""
    if t.is_cuda and _wavelets._device_taps_mode != "never" and (device is None or t.device == device):
        return _engine.DevTaps(t)
    return [float(v) for v in t.detach().double().cpu().reshape(-1).tolist()]


class _Axis1(torch.autograd.Function):
    """rows [R, N] -> [R, 2, M]: one analysis level along the last axis, taps as tensors."""

    @staticmethod
    def forward(ctx, x, lo_t, hi_t, mode_id):
        ctx.mode_id = mode_id
        ctx.save_for_backward(x, lo_t, hi_t)
        return _engine.ENGINE.analysis(x, _host_taps_of(lo_t, x.device), _host_taps_of(hi_t, x.device), mode_id)

    @staticmethod
    def backward(ctx, g):
        x, lo_t, hi_t = ctx.saved_tensors
        g_x = _Axis1Adj.apply(g, lo_t, hi_t, ctx.mode_id, x.shape[1]) if ctx.needs_input_grad[0] else None
        t_lo = t_hi = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            t_lo, t_hi = _Axis1Corr.apply(x, g, lo_t.numel(), ctx.mode_id)
            t_lo, t_hi = _like(t_lo, lo_t), _like(t_hi, hi_t)
        return g_x, t_lo, t_hi, None


class _Axis1Adj(torch.autograd.Function):
    """[R, 2, M] -> rows [R, N]: the transpose of :class:`_Axis1`."""

    @staticmethod
    def forward(ctx, g, lo_t, hi_t, mode_id, n):
        ctx.mode_id = mode_id
        ctx.save_for_backward(g, lo_t, hi_t)
        return _engine.ENGINE.analysis_adjoint(g, (n,), _host_taps_of(lo_t, g.device), _host_taps_of(hi_t, g.device), mode_id)

    @staticmethod
    def backward(ctx, gg):
        g, lo_t, hi_t = ctx.saved_tensors
        g_g = _Axis1.apply(gg, lo_t, hi_t, ctx.mode_id) if ctx.needs_input_grad[0] else None
        t_lo = t_hi = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:  # <gg, A(h)^T g> = <A(h) gg, g>
            t_lo, t_hi = _Axis1Corr.apply(gg, g, lo_t.numel(), ctx.mode_id)
            t_lo, t_hi = _like(t_lo, lo_t), _like(t_hi, hi_t)
        return g_g, t_lo, t_hi, None, None


class _Axis1Corr(torch.autograd.Function):
    """(rows x [R, N], g [R, 2, M]) -> (t_lo, t_hi) float64 [L]:  t_b[m] = sum_{row, k} g_b[row, k] x_ext[row, 2 k + 1 - m]."""

    @staticmethod
    def forward(ctx, x, g, flen, mode_id):
        ctx.mode_id = mode_id
        ctx.save_for_backward(x, g)
        t_lo = torch.zeros(flen, dtype=torch.float64, device=x.device)
        t_hi = torch.zeros_like(t_lo)
        _engine.ENGINE.tap_correlate(g[:, 0], x, flen, 1, -1, mode_id, t_lo)
        _engine.ENGINE.tap_correlate(g[:, 1], x, flen, 1, -1, mode_id, t_hi)
        return t_lo, t_hi

    @staticmethod
    def backward(ctx, w_lo, w_hi):
        x, g = ctx.saved_tensors
        w_lo, w_hi = w_lo.to(x.dtype), w_hi.to(x.dtype)
        g_x = _Axis1Adj.apply(g, w_lo, w_hi, ctx.mode_id, x.shape[1]) if ctx.needs_input_grad[0] else None
        g_g = _Axis1.apply(x, w_lo, w_hi, ctx.mode_id) if ctx.needs_input_grad[1] else None
        return g_x, g_g, None, None


class _Syn1(torch.autograd.Function):
    """(a, d [R, M]) -> y [R, n_out]: one synthesis level along the last axis (cropped to n_out), taps as tensors."""

    @staticmethod
    def forward(ctx, a, d, lo_t, hi_t, n_out):
        ctx.save_for_backward(a, d, lo_t, hi_t)
        return _engine.ENGINE.synthesis(a, [d], _host_taps_of(lo_t, a.device), _host_taps_of(hi_t, a.device), [n_out])

    @staticmethod
    def backward(ctx, g_y):
        a, d, lo_t, hi_t = ctx.saved_tensors
        g_a = g_d = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            gb = _Syn1Adj.apply(g_y, lo_t, hi_t, a.shape[1])
            g_a, g_d = gb[:, 0], gb[:, 1]
        t_lo = t_hi = None
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
            t_lo, t_hi = _Syn1Corr.apply(a, d, g_y, lo_t.numel())
            t_lo, t_hi = _like(t_lo, lo_t), _like(t_hi, hi_t)
        return g_a, g_d, t_lo, t_hi, None


class _Syn1Adj(torch.autograd.Function):
    """g_y [R, n_out] -> [R, 2, M]: the transpose of :class:`_Syn1`."""

    @staticmethod
    def forward(ctx, g_y, lo_t, hi_t, m):
        ctx.save_for_backward(g_y, lo_t, hi_t)
        return _engine.ENGINE.synthesis_adjoint(g_y, (m,), _host_taps_of(lo_t, g_y.device), _host_taps_of(hi_t, g_y.device))

    @staticmethod
    def backward(ctx, gg):
        g_y, lo_t, hi_t = ctx.saved_tensors
        g_gy = _Syn1.apply(gg[:, 0], gg[:, 1], lo_t, hi_t, g_y.shape[1]) if ctx.needs_input_grad[0] else None
        t_lo = t_hi = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:  # <gg, S(r)^T g_y> = <S(r) gg, g_y>
            t_lo, t_hi = _Syn1Corr.apply(gg[:, 0], gg[:, 1], g_y, lo_t.numel())
            t_lo, t_hi = _like(t_lo, lo_t), _like(t_hi, hi_t)
        return g_gy, t_lo, t_hi, None


class _Syn1Corr(torch.autograd.Function):
    """(a, d [R, M], g_y [R, n_out]) -> (t_lo, t_hi) float64 [L]:  t_lo[j] = sum_{row, k} a[row, k] g_y[row, 2 k - (L - 2) + j] (zeros outside)."""

    @staticmethod
    def forward(ctx, a, d, g_y, flen):
        ctx.save_for_backward(a, d, g_y)
        t_lo = torch.zeros(flen, dtype=torch.float64, device=a.device)
        t_hi = torch.zeros_like(t_lo)
        _engine.ENGINE.tap_correlate(a, g_y, flen, -(flen - 2), 1, 0, t_lo)
        _engine.ENGINE.tap_correlate(d, g_y, flen, -(flen - 2), 1, 0, t_hi)
        return t_lo, t_hi

    @staticmethod
    def backward(ctx, w_lo, w_hi):
        a, d, g_y = ctx.saved_tensors
        w_lo, w_hi = w_lo.to(a.dtype), w_hi.to(a.dtype)
        g_a = g_d = g_gy = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            gb = _Syn1Adj.apply(g_y, w_lo, w_hi, a.shape[1])
            g_a, g_d = gb[:, 0], gb[:, 1]
        if ctx.needs_input_grad[2]:
            g_gy = _Syn1.apply(a, d, w_lo, w_hi, g_y.shape[1])
        return g_a, g_d, g_gy, None


def _analysis_level_closed(x: torch.Tensor, lo_t: torch.Tensor, hi_t: torch.Tensor, mode_id: int) -> torch.Tensor:
    """One N-D analysis level [B, N..] -> [B, 2^n, M..] as a product of :class:`_Axis1` ops (last axis first): differentiable to any
    order w.r.t. the data and the taps."""
    nd = x.dim() - 1
    bands = {0: x}
    for a in reversed(range(nd)):
        bit = 1 << (nd - 1 - a)
        nxt = {}
        for s, t in bands.items():
            tt = t.movedim(1 + a, -1)
            out = _Axis1.apply(tt.reshape(-1, tt.shape[-1]), lo_t, hi_t, mode_id)  # [rows, 2, M]
            for sigma in (0, 1):
                nxt[s | (bit * sigma)] = out[:, sigma].reshape(*tt.shape[:-1], out.shape[-1]).movedim(-1, 1 + a)
        bands = nxt
    return torch.stack([bands[s] for s in range(1 << nd)], dim=1)


def _synthesis_level_closed(approx: torch.Tensor, details: Sequence[torch.Tensor], lo_t: torch.Tensor, hi_t: torch.Tensor,
                            out_ext: Sequence[int]) -> torch.Tensor:
    """One N-D synthesis level (bands [B, M..] in band order -> [B, *out_ext]) as a product of :class:`_Syn1` ops (first axis first)."""
    nd = approx.dim() - 1
    bands = {s: t for s, t in enumerate([approx, *details])}
    for a in range(nd):
        bit = 1 << (nd - 1 - a)
        nxt = {}
        for s, t in bands.items():
            if s & bit:
                continue
            lo, hi = t.movedim(1 + a, -1), bands[s | bit].movedim(1 + a, -1)
            y = _Syn1.apply(lo.reshape(-1, lo.shape[-1]), hi.reshape(-1, hi.shape[-1]), lo_t, hi_t, int(out_ext[a]))
            nxt[s] = y.reshape(*lo.shape[:-1], y.shape[-1]).movedim(-1, 1 + a)
        bands = nxt
    return bands[0]


def _partials_at(values, build, cotangent, weights):
    """Second-order partial derivatives by automatic differentiation AT DETACHED COPIES: ``values`` are the op's inputs, ``build``
    maps fresh leaves holding their values to the level's output, ``cotangent`` (one of the leaves' positions, see callers) is the
    upstream gradient.  Returns d/d(leaf) of  sum_i <weights_i, d <output, cotangent> / d leaf_i>  for every leaf: the derivatives of
    the first-order gradients (as a vector-Jacobian product with ``weights``) w.r.t. every input of the op, each a PARTIAL derivative —
    the leaves have no history, so nothing upstream is counted twice."""
    leaves = [v.detach().clone().requires_grad_(True) for v in values]
    with torch.enable_grad():
        out = build(leaves)
        first = torch.autograd.grad(out, leaves[1:], leaves[0], create_graph=True, allow_unused=True)  # (leaves[0] = the cotangent)
        s = None
        for f, w in zip(first, weights):
            if f is not None and w is not None:
                term = (f * w.to(f.dtype)).sum()
                s = term if s is None else s + term
        if s is None:
            return [None] * len(leaves)
        return list(torch.autograd.grad(s, leaves, allow_unused=True))


def _third_order_refused(taps) -> None:
    if torch.is_grad_enabled() and any(t.requires_grad for t in taps):
        raise RuntimeError("ptwt_amd: derivatives beyond second order through a learnable filter bank are not supported "
                           "(create_graph=True inside a double backward); first and second order are.")


class _AnalysisLevelGrad(torch.autograd.Function):
    """(g_buf, x, dec taps) -> (g_x, g_lo, g_hi): the first-order gradients of one analysis level, from the fused kernels as in
    :class:`_AnalysisLevel`, as an op of its own so that a double backward (create_graph=True) with a LEARNABLE filter bank gets the
    mixed second derivatives: its backward differentiates the level — rebuilt from the per-axis ops that are closed under
    differentiation (:func:`_analysis_level_closed`) — twice at detached copies of the inputs.  The reference has these terms from
    ATen's conv / pad backward (src/ptwt/_util.py:115-132 keeps the taps in the graph)."""

    @staticmethod
    def forward(ctx, g_buf, x, lo_t, hi_t, dec_lo, dec_hi, mode_id):
        ctx.mode_id = mode_id
        ctx.save_for_backward(g_buf, x, lo_t, hi_t)
        g_x = _engine.ENGINE.analysis_adjoint(g_buf, tuple(x.shape[1:]), dec_lo, dec_hi, mode_id)
        g_lo, g_hi = _analysis_tap_grads(x, g_buf, dec_lo, dec_hi, mode_id)
        return g_x, _like(g_lo, lo_t), _like(g_hi, hi_t)

    @staticmethod
    def backward(ctx, w_x, w_lo, w_hi):
        g_buf, x, lo_t, hi_t = ctx.saved_tensors
        _third_order_refused((lo_t, hi_t))
        mode_id = ctx.mode_id
        d = _partials_at([g_buf, x, lo_t, hi_t], lambda lv: _analysis_level_closed(lv[1], lv[2], lv[3], mode_id), 0, [w_x, w_lo, w_hi])
        return d[0], d[1], d[2], d[3], None, None, None


class _SynthesisLevelGrad(torch.autograd.Function):
    """(g_y, rec taps, approx, details) -> (g_lo, g_hi, g_approx, g_details...): the first-order gradients of one synthesis level (fused
    kernels) as an op whose backward has the mixed second derivatives with a learnable filter bank (see :class:`_AnalysisLevelGrad`)."""

    @staticmethod
    def forward(ctx, g_y, lo_t, hi_t, rec_lo, rec_hi, approx, *details):
        ctx.save_for_backward(g_y, lo_t, hi_t, approx, *details)
        g = _engine.ENGINE.synthesis_adjoint(g_y, tuple(approx.shape[1:]), rec_lo, rec_hi)
        g_lo, g_hi = _synthesis_tap_grads(g_y, approx, details, rec_lo, rec_hi)
        return (_like(g_lo, lo_t), _like(g_hi, hi_t), *[g[:, s] for s in range(len(details) + 1)])

    @staticmethod
    def backward(ctx, w_lo, w_hi, *w_bands):
        g_y, lo_t, hi_t, approx, *details = ctx.saved_tensors
        _third_order_refused((lo_t, hi_t))
        out_ext = tuple(g_y.shape[1:])
        d = _partials_at([g_y, lo_t, hi_t, approx, *details], lambda lv: _synthesis_level_closed(lv[3], lv[4:], lv[1], lv[2], out_ext), 0,
                         [w_lo, w_hi, *w_bands])
        return (d[0], d[1], d[2], None, None, *d[3:])


def _like(grad64: torch.Tensor, ref: Optional[torch.Tensor]):
    return None if ref is None else grad64.to(device=ref.device, dtype=ref.dtype).reshape(ref.shape)


class _AnalysisLevel(torch.autograd.Function):
    """One analysis level as a differentiable op w.r.t. its input and (optionally) the dec taps.  The reference is
    differentiable because it is built from ATen ops (F.pad + F.conv*d, src/ptwt/conv_transform.py:135-139 and the
    2-D / 3-D twins); here the backward is the explicit adjoint kernel (C ABI ``mifwt_dwt_fwd_adjoint``) and, for
    learnable filter banks, the tap correlation (``mifwt_tap_correlate``).  ``lo_t`` / ``hi_t`` are the tap TENSORS (or
    None): they only tie the op into the graph, the kernels take the host copies ``dec_lo`` / ``dec_hi``."""

    @staticmethod
    def forward(ctx, x, dec_lo, dec_hi, mode_id, lo_t=None, hi_t=None):
        ctx.meta = (tuple(x.shape[1:]), dec_lo, dec_hi, mode_id)
        ctx.taps = (lo_t, hi_t)
        need_taps = any(t is not None and t.requires_grad for t in (lo_t, hi_t))
        ctx.save_for_backward(x if need_taps else None)
        return _engine.ENGINE.analysis(x, dec_lo, dec_hi, mode_id)

    @staticmethod
    def backward(ctx, g_buf):
        sig_shape, dec_lo, dec_hi, mode_id = ctx.meta
        (x,) = ctx.saved_tensors
        if x is not None and torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ctx.taps):
            # a graph of this backward is wanted (create_graph=True) and the filter bank is learnable: the same first-order gradients as
            # an op whose own backward has the mixed second derivatives (data x taps, taps x taps, upstream gradient x taps)
            lo_t, hi_t = ctx.taps
            g_x, g_lo, g_hi = _AnalysisLevelGrad.apply(g_buf, x, lo_t, hi_t, dec_lo, dec_hi, mode_id)
            return (g_x if ctx.needs_input_grad[0] else None, None, None, None,
                    g_lo if ctx.needs_input_grad[4] else None, g_hi if ctx.needs_input_grad[5] else None)
        # the adjoint is itself a differentiable level op (its derivative is this level again): gradients of any order w.r.t.
        # the data, as the reference has them from plain F.pad + conv.  (The tap gradients below are first order only.)
        g_x = _AnalysisAdjointLevel.apply(g_buf, sig_shape, dec_lo, dec_hi, mode_id) if ctx.needs_input_grad[0] else None
        g_lo = g_hi = None
        if x is not None and (ctx.needs_input_grad[4] or ctx.needs_input_grad[5]):
            g_lo, g_hi = _analysis_tap_grads(x, g_buf.detach(), dec_lo, dec_hi, mode_id)
            g_lo, g_hi = _like(g_lo, ctx.taps[0]), _like(g_hi, ctx.taps[1])
        return g_x, None, None, None, g_lo, g_hi


class _AnalysisAdjointLevel(torch.autograd.Function):
    """The transpose of one analysis level, ``g_buf`` [B, 2^n, M..] -> ``g_x`` [B, N..] (C ABI ``mifwt_dwt_fwd_adjoint``) — linear, so
    its own backward is the analysis level."""

    @staticmethod
    def forward(ctx, g_buf, sig_shape, dec_lo, dec_hi, mode_id):
        ctx.meta = (dec_lo, dec_hi, mode_id)
        return _engine.ENGINE.analysis_adjoint(g_buf, sig_shape, dec_lo, dec_hi, mode_id)

    @staticmethod
    def backward(ctx, gg_x):
        dec_lo, dec_hi, mode_id = ctx.meta
        return _AnalysisLevel.apply(gg_x, dec_lo, dec_hi, mode_id, None, None), None, None, None, None


class _AnalysisAdjointBands(torch.autograd.Function):
    """:class:`_AnalysisAdjointLevel` with the gradient of every band in a tensor of its own (no concatenation: the C ABI takes a
    pointer per band); linear as well, its backward is the analysis level."""

    @staticmethod
    def forward(ctx, sig_shape, dec_lo, dec_hi, mode_id, g_approx, *g_details):
        ctx.meta = (dec_lo, dec_hi, mode_id, len(g_details))
        return _engine.ENGINE.analysis_adjoint_bands(g_approx, g_details, sig_shape, dec_lo, dec_hi, mode_id)

    @staticmethod
    def backward(ctx, gg_x):
        dec_lo, dec_hi, mode_id, ndet = ctx.meta
        buf = _AnalysisLevel.apply(gg_x, dec_lo, dec_hi, mode_id, None, None)
        return (None, None, None, None) + tuple(buf[:, s] for s in range(ndet + 1))


class _AnalysisPyramid(torch.autograd.Function):
    """Several 2-D analysis levels in ONE launch as a differentiable op w.r.t. the data (C ABI ``mifwt_dwt2_fwd_pyramid``: the streaming
    three-level kernel / the small-plane kernel; src/ptwt/conv_transform_2.py:142-149 per trip): the forward of a call that asks for
    gradients runs on the same fused kernels as one that does not, and saves nothing.  The outputs are the BANDS (views of the level
    buffers): the approximation of the last fused level, then (ad, da, dd) per level, finest first — autograd hands the backward one
    gradient per band, no buffer is assembled.  The backward composes the per-level adjoints coarse to fine (each a differentiable op in
    turn: gradients of any order)."""

    @staticmethod
    def forward(ctx, x, dec_lo, dec_hi, mode_id, want):
        bufs = _engine.ENGINE.analysis_pyramid(x, dec_lo, dec_hi, mode_id, want)
        shapes, n = [], list(x.shape[1:])
        out = [bufs[-1][:, 0]]
        for b in bufs:
            shapes.append(tuple(n))
            n = list(b.shape[2:])
            out.extend(b.unbind(1)[-3:])
        ctx.meta = (shapes, dec_lo, dec_hi, mode_id)
        return tuple(out)

    @staticmethod
    def backward(ctx, g_approx, *g_bands):
        shapes, dec_lo, dec_hi, mode_id = ctx.meta
        if mode_id == _engine.MODE_IDS["zero"] and not torch.is_grad_enabled():
            # zero mode, no graph of the backward wanted: the adjoint of every level is a synthesis level with the dec taps reversed,
            # so the adjoint of the whole launch is ONE multi-level synthesis launch (kernels 22 / 21) where the library serves it
            levels = [list(g_bands[3 * lvl : 3 * lvl + 3]) for lvl in range(len(shapes) - 1, -1, -1)]  # coarsest first
            g = _engine.ENGINE.synthesis_pyramid(g_approx if g_approx.stride(-1) == 1 else g_approx.contiguous(), levels, dec_lo[::-1], dec_hi[::-1],
                                                 list(shapes[0]))
            if g is not None:
                return g, None, None, None, None
        g = g_approx
        for lvl in range(len(shapes) - 1, -1, -1):
            g = _AnalysisAdjointBands.apply(shapes[lvl], dec_lo, dec_hi, mode_id, g, *g_bands[3 * lvl : 3 * lvl + 3])
        return g, None, None, None, None


class _FusedRouteUnavailable(Exception):
    """Raised inside the forward of a multi-level autograd op when the library turns the launch down after all; the caller goes level
    by level."""


class _AnalysisTail(torch.autograd.Function):
    """Several 1-D analysis levels in one launch (C ABI ``mifwt_dwt1_fwd_long`` / ``mifwt_dwt1_fwd_tail``; src/ptwt/conv_transform.py:133-140
    per trip) as a differentiable op w.r.t. the data: outputs = the last level's approximation, then the detail rows, finest first; the
    backward composes the per-level adjoints coarse to fine."""

    @staticmethod
    def forward(ctx, x, dec_lo, dec_hi, mode_id, want):
        bufs = _engine.ENGINE.analysis_tail(x, dec_lo, dec_hi, mode_id, want)
        if bufs is None:
            raise _FusedRouteUnavailable()
        shapes, n = [], int(x.shape[1])
        for b in bufs:
            shapes.append((n,))
            n = int(b.shape[2])
        ctx.meta = (shapes, dec_lo, dec_hi, mode_id)
        return (bufs[-1][:, 0], *[b[:, -1] for b in bufs])

    @staticmethod
    def backward(ctx, g_approx, *g_details):
        shapes, dec_lo, dec_hi, mode_id = ctx.meta
        g = g_approx
        for lvl in range(len(shapes) - 1, -1, -1):
            g = _AnalysisAdjointBands.apply(shapes[lvl], dec_lo, dec_hi, mode_id, g, g_details[lvl])
        return g, None, None, None, None


class _SynthesisChain1d(torch.autograd.Function):
    """A whole 1-D reconstruction on the multi-level launches (``mifwt_dwt1_inv_tail`` for the coarse levels, ``mifwt_dwt1_inv_long`` for
    the fine ones; src/ptwt/conv_transform.py:184-199 per trip) as a differentiable op w.r.t. the coefficients; ``run`` does the launches
    (the caller's recursion over the two kernels), the backward composes the per-level synthesis adjoints fine to coarse."""

    @staticmethod
    def forward(ctx, run, rec_lo, rec_hi, approx, *dets):
        ctx.meta = (rec_lo, rec_hi, [tuple(d.shape[1:]) for d in dets])
        return run(approx, list(dets))

    @staticmethod
    def backward(ctx, g_y):
        rec_lo, rec_hi, coef_shapes = ctx.meta
        grads: list = []
        g = g_y
        todo = list(reversed(coef_shapes))  # finest level first
        # no graph of the backward wanted: the adjoint of the chain is a zero-mode multi-level ANALYSIS with the rec taps reversed (a
        # trimmed output sample is a zero of the zero extension) — the 1-D multi-level launches (kernels 17 / 14), see _SynthesisPyramid
        fused = not torch.is_grad_enabled() and not _engine._is_dev(rec_lo) and g_y.dim() == 2
        zero = _engine.MODE_IDS["zero"]
        flen = len(rec_lo)
        while todo:
            bufs = None
            if fused and len(todo) >= 2:
                # what a zero-mode analysis of g yields level by level (floor((n + L - 1) / 2)) against the coefficient shapes the forward
                # took: decided on the host BEFORE anything is launched — a chain with a crop the adjoint launch cannot express (a
                # separable crop of more than one sample) goes level by level from here on instead of launching, discarding and retrying
                n, ok = int(g.shape[-1]), 0
                for shp in todo:
                    n = (n + flen - 1) // 2
                    if (n,) != tuple(shp):
                        break
                    ok += 1
                if ok < 2:
                    fused = False
                else:
                    bufs = _engine.ENGINE.analysis_tail(g if g.stride(-1) == 1 else g.contiguous(), list(rec_lo)[::-1], list(rec_hi)[::-1], zero, ok)
                    if bufs is not None and (len(bufs) < 2 or any(tuple(b.shape[2:]) != tuple(shp) for b, shp in zip(bufs, todo))):
                        bufs, fused = None, False
            if bufs is None:
                gb = _SynthesisAdjointLevel.apply(g, todo[0], rec_lo, rec_hi)
                grads.insert(0, gb[:, 1])
                g = gb[:, 0]
                todo = todo[1:]
            else:
                for b in bufs:
                    grads.insert(0, b[:, -1])
                g = bufs[-1][:, 0]
                todo = todo[len(bufs):]
        return (None, None, None, g, *grads)


class _SynthesisLevel(torch.autograd.Function):
    """One synthesis level, differentiable w.r.t. the approximation, every detail band and (optionally) the rec taps
    (backward: ``mifwt_dwt_inv_adjoint`` + ``mifwt_tap_correlate``; reference: autograd through torch.stack +
    F.conv_transpose*d + crop)."""

    @staticmethod
    def forward(ctx, rec_lo, rec_hi, out_ext, lo_t, hi_t, approx, *details):
        ctx.meta = (tuple(approx.shape[1:]), rec_lo, rec_hi, len(details))
        ctx.taps = (lo_t, hi_t)
        need_taps = any(t is not None and t.requires_grad for t in (lo_t, hi_t))
        if need_taps:
            ctx.save_for_backward(approx, *details)
        else:
            ctx.save_for_backward()
        return _engine.ENGINE.synthesis(approx, list(details), rec_lo, rec_hi, out_ext)

    @staticmethod
    def backward(ctx, g_y):
        coef_shape, rec_lo, rec_hi, ndet = ctx.meta
        saved = ctx.saved_tensors
        if saved and torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ctx.taps):
            # create_graph=True with a learnable filter bank: first-order gradients as an op that carries the mixed second derivatives
            lo_t, hi_t = ctx.taps
            out = _SynthesisLevelGrad.apply(g_y, lo_t, hi_t, rec_lo, rec_hi, *saved)
            need = ctx.needs_input_grad
            return (None, None, None, out[0] if need[3] else None, out[1] if need[4] else None,
                    *[o if need[5 + i] else None for i, o in enumerate(out[2:])])
        g_bands = (None,) * (ndet + 1)
        if any(ctx.needs_input_grad[5:]):
            g = _SynthesisAdjointLevel.apply(g_y, coef_shape, rec_lo, rec_hi)  # differentiable in turn: any order w.r.t. the data
            g_bands = tuple(g[:, s] for s in range(ndet + 1))
        g_lo = g_hi = None
        if saved and (ctx.needs_input_grad[3] or ctx.needs_input_grad[4]):
            g_lo, g_hi = _synthesis_tap_grads(g_y.detach(), saved[0], saved[1:], rec_lo, rec_hi)
            g_lo, g_hi = _like(g_lo, ctx.taps[0]), _like(g_hi, ctx.taps[1])
        return (None, None, None, g_lo, g_hi) + g_bands


class _SynthesisAdjointLevel(torch.autograd.Function):
    """The transpose of one synthesis level, ``g_y`` [B, out..] -> [B, 2^n, M..] (C ABI ``mifwt_dwt_inv_adjoint``); its own backward is
    the synthesis level."""

    @staticmethod
    def forward(ctx, g_y, coef_shape, rec_lo, rec_hi):
        ctx.meta = (tuple(g_y.shape[1:]), rec_lo, rec_hi)
        return _engine.ENGINE.synthesis_adjoint(g_y, coef_shape, rec_lo, rec_hi)

    @staticmethod
    def backward(ctx, gg):
        out_ext, rec_lo, rec_hi = ctx.meta
        bands = [gg[:, s] for s in range(gg.shape[1])]
        return _SynthesisLevel.apply(rec_lo, rec_hi, out_ext, None, None, bands[0], *bands[1:]), None, None, None


class _SynthesisPyramid(torch.autograd.Function):
    """Several 2-D synthesis levels in ONE launch as a differentiable op w.r.t. the coefficients (C ABI ``mifwt_dwt2_inv_pyramid``: the
    streaming three-level kernel / the small-plane kernel; src/ptwt/conv_transform_2.py:222-249 per trip): a `waverec2` that asks for
    gradients runs its forward on the same fused kernels as one that does not.  ``dets`` = three detail bands per level, coarsest level
    first.  The backward composes the per-level synthesis adjoints (zero-mode analysis kernels) fine to coarse, each a differentiable
    op in turn."""

    @staticmethod
    def forward(ctx, rec_lo, rec_hi, out_ext, plan, nlev, approx, *dets):
        levels = [list(dets[3 * l : 3 * l + 3]) for l in range(nlev)]
        ctx.meta = (rec_lo, rec_hi, [tuple(lv[0].shape[1:]) for lv in levels])
        y = _engine.ENGINE.synthesis_pyramid(approx, levels, rec_lo, rec_hi, out_ext, plan=plan)
        if y is None:  # (bands of a level that do not share their strides: level by level, same op)
            y = approx
            for k, lv in enumerate(levels):
                nxt = levels[k + 1][0].shape[1:] if k + 1 < nlev else out_ext
                y = _engine.ENGINE.synthesis(y, lv, rec_lo, rec_hi, list(nxt))
        return y

    @staticmethod
    def backward(ctx, g_y):
        rec_lo, rec_hi, coef_shapes = ctx.meta
        grads: list = []
        g = g_y
        todo = list(reversed(coef_shapes))  # finest level first: its adjoint yields the gradient of its four bands
        # No graph of the backward wanted: the adjoint of a synthesis level is a zero-mode analysis level with the rec taps reversed, and a
        # trimmed output row / column is a zero the zero extension supplies anyway — so the adjoint of the whole reconstruction is ONE
        # multi-level analysis launch (kernels 16 / 20) where the library serves it (the mirror of _AnalysisPyramid.backward).
        fused = not torch.is_grad_enabled() and not _engine._is_dev(rec_lo) and g_y.dim() == 3 and g_y.dtype == torch.float32
        zero = _engine.MODE_IDS["zero"]
        while todo:
            bufs = None
            if fused and len(todo) >= 2:
                gin = g if g.stride(-1) == 1 else g.contiguous()
                # the zero-mode analysis extents (floor((n + L - 1) / 2) per axis and level) against the coefficient shapes of the forward,
                # on the host, before anything is launched: a crop the adjoint launch cannot express sends the rest level by level
                ns, ok = [int(v) for v in gin.shape[1:]], 0
                for shp in todo:
                    ns = [(n + len(rec_lo) - 1) // 2 for n in ns]
                    if tuple(ns) != tuple(shp):
                        break
                    ok += 1
                k = _engine.ENGINE.pyramid_levels(gin, len(rec_lo), zero, ok) if ok >= 2 else 0  # (a query: nothing is launched)
                if ok < 2:
                    fused = False
                if k >= 2:
                    bufs = _engine.ENGINE.analysis_pyramid(gin, list(rec_lo)[::-1], list(rec_hi)[::-1], zero, k)
                if bufs is not None and (len(bufs) < 2 or any(tuple(b.shape[2:]) != tuple(shp) for b, shp in zip(bufs, todo))):
                    bufs, fused = None, False
            if bufs is None:
                gb = _SynthesisAdjointLevel.apply(g, todo[0], rec_lo, rec_hi)
                grads = [gb[:, 1], gb[:, 2], gb[:, 3]] + grads
                g = gb[:, 0]  # = the gradient of the coarser level's (trimmed) output
                todo = todo[1:]
            else:
                for b in bufs:
                    grads = list(b.unbind(1)[-3:]) + grads
                g = bufs[-1][:, 0]
                todo = todo[len(bufs):]
        return (None, None, None, None, None, g, *grads)


def _tap_tensors(wavelet):
    """(dec_lo, dec_hi, rec_lo, rec_hi) as the caller's TENSORS when the filter bank is learnable (any of them requires
    grad and grad mode is on), else None.  src/ptwt/_util.py:115-132 keeps such taps in the graph with torch.as_tensor."""
    if isinstance(wavelet, str) or not torch.is_grad_enabled():
        return None
    bank = wavelet if isinstance(wavelet, tuple) else getattr(wavelet, "filter_bank", ())
    if len(bank) == 4 and all(isinstance(t, torch.Tensor) for t in bank) and any(t.requires_grad for t in bank):
        return tuple(bank)
    return None


# detail-band order of each public container, as band indices of the engine (bit (n-1-a) <=> axis a high-pass)
_KEYS_ND = {
    2: ("ad", "da", "dd"),
    3: ("aad", "ada", "add", "daa", "dad", "dda", "ddd"),  # src/ptwt/conv_transform_3.py:131-141
}
# insertion order of the separable recursion (src/ptwt/separable_conv_transform.py:63-72)
_KEYS_FS = {
    2: ("da", "ad", "dd"),
    3: ("daa", "ada", "dda", "aad", "dad", "add", "ddd"),
}


def _band(key: str) -> int:
    return int(key.replace("a", "0").replace("d", "1"), 2)


# ------------------------------------------------------------------------------------------ arguments
def _mode_id(mode) -> int:
    """src/ptwt/_util.py:48-68: None means reflect, unknown strings are a ValueError."""
    if mode is None:
        mode = "reflect"
    try:
        return _engine.MODE_IDS[mode]
    except (KeyError, TypeError):
        raise ValueError(f"Padding mode not supported: {mode}") from None


def _ensure_axes(axes: AxisHint, ndim: int) -> Tuple[int, ...]:
    """src/ptwt/_util.py:817-826."""
    if axes is None:
        return tuple(range(-ndim, 0))
    if isinstance(axes, int):
        if ndim != 1:
            raise ValueError(f"tried passing single axis to {ndim}D transform")
        return (axes,)
    if len(axes) != ndim:
        raise ValueError(f"tried passing {len(axes)}D axes {axes} to {ndim}D transform")
    if len(set(axes)) != len(axes):
        raise ValueError("Cant transform the same axis twice.")
    return tuple(axes)


def _permutation(axes: Sequence[int], rank: int) -> List[int]:
    """Order that moves ``axes`` (in the given order) to the end (src/ptwt/_util.py:351-363)."""
    back = [a + rank if a < 0 else a for a in axes]
    if len(set(back)) != len(back):
        raise ValueError("Cant transform the same axis twice.")
    if any(a < 0 or a >= rank for a in back):
        raise ValueError(f"axes {tuple(axes)} out of range for a {rank}-dimensional tensor")
    return [a for a in range(rank) if a not in back] + back


class _Layout:
    """Remembers how a tensor was brought to [B, N_0..N_{n-1}] so that results can be taken back
    (src/ptwt/_util.py:493-570 forward, :613-676 backward)."""

    def __init__(self, proto: torch.Tensor, ndim: int, axes: Tuple[int, ...]):
        if not isinstance(proto, torch.Tensor):
            raise ValueError("First element of coeffs must be the approximation coefficient tensor.")
        if proto.dtype not in supported_dtypes():
            raise ValueError(f"Input dtype {proto.dtype} not supported")
        self.ndim = ndim
        self.perm = None if axes == tuple(range(-ndim, 0)) else _permutation(axes, proto.dim())
        shape = list(proto.shape) if self.perm is None else [proto.shape[p] for p in self.perm]
        if len(shape) < ndim:
            raise ValueError(f"At least {ndim} input dimensions required.")
        self.lead = shape[:-ndim]  # leading (batch) dims after the swap

    def fold(self, t: torch.Tensor) -> torch.Tensor:
        if self.perm is not None:
            t = t.permute(self.perm)
        if len(self.lead) == 0:
            return t.unsqueeze(0)
        if len(self.lead) > 1:
            return t.reshape([-1] + list(t.shape[-self.ndim:]))
        return t

    def unfold(self, t: torch.Tensor) -> torch.Tensor:
        if len(self.lead) == 0:
            t = t.squeeze(0)
        elif len(self.lead) > 1:
            t = t.reshape(self.lead + list(t.shape[-self.ndim:]))
        if self.perm is not None:
            inv = [0] * len(self.perm)
            for i, p in enumerate(self.perm):
                inv[p] = i
            t = t.permute(inv)
        return t


def _check_pad(extents: Sequence[int], flen: int, mode: str) -> None:
    """The reference pads with torch.nn.functional.pad, which refuses reflect pad >= N and circular
    pad > N (RuntimeError); symmetric/zero/constant accept anything (src/ptwt/conv_transform.py:59-66)."""
    if mode not in ("reflect", "periodic"):
        return
    for n in extents:
        padl = (2 * flen - 3) // 2
        padr = padl + n % 2
        if mode == "reflect" and max(padl, padr) >= n:
            raise RuntimeError(
                f"Padding size should be less than the corresponding input dimension, but got: padding "
                f"({padl}, {padr}) at an axis of extent {n} (reflect mode, filter length {flen})"
            )
        if mode == "periodic" and max(padl, padr) > n:
            raise RuntimeError(
                f"Padding value causes wrapping around more than once: padding ({padl}, {padr}) at an axis "
                f"of extent {n} (periodic mode, filter length {flen})"
            )


# ------------------------------------------------------------------------------------------ analysis
# geometry of a graph-free decomposition -> the launches it took ((0, levels asked of the multi-level launch) / (1, 0) level pair /
# (2, levels) 1-D tail / (3, 0) one level); idempotent values, single dict operations (see the synthesis memos below)
_route_memo: dict = {}
_engine._routing_caches.append(_route_memo)


def analysis(data: torch.Tensor, wavelet, mode, level: Optional[int], axes: AxisHint, ndim: int):
    """Multi-level analysis.  Returns ``(layout, approx [B,*M], [buffer [B,2^n,*M] per level, coarsest first])``."""
    axes = _ensure_axes(axes, ndim)
    layout = _Layout(data, ndim, axes)
    x = layout.fold(data)
    dbank = device_bank(wavelet, x.device) if x.is_cuda else None
    if dbank is not None:  # the taps stay on the GPU: per-level generic route, nothing read back (include/mifwt.h, mifwt_*_dtaps)
        dec_lo, dec_hi = _engine.DevTaps(dbank[0]), _engine.DevTaps(dbank[1])
    else:
        dec_lo, dec_hi, _, _ = host_taps(wavelet)
    on_device = dbank is not None
    tap_t = _tap_tensors(wavelet)
    flen = len(dec_lo)
    if level is None:
        level = dwtn_max_level(x.shape[1:], flen)
    bufs: List[torch.Tensor] = []
    cur = x
    done = 0
    # Calls that need no graph (inference: the common case, and every timed loop) REPLAY the launches their geometry took last time:
    # which levels fuse is a function of the geometry and the routing options alone, and the pad checks below raise by geometry alone —
    # a geometry is memoised only after it passed them.  (32 x 1000^2 db5 periodic level 5, five launches: 79 -> ~60 us of host time a
    # call, which is what that call costs once the GPU needs less: tools/host_profile_ref.py)
    rkey = None
    if not on_device and not (torch.is_grad_enabled() and (x.requires_grad or tap_t is not None)):
        mode_id = _mode_id(mode)
        eng = _engine.ENGINE
        rkey = (ndim, x.shape, x.stride(), x.dtype, x.device, flen, mode_id, level, id(eng), _engine.MAX_PYRAMID_LEVELS, _engine.ROW_ALIGN,
                _engine.PYRAMID_ROW_ALIGN)
        steps = _route_memo.get(rkey)
        if steps is not None:
            for kind, arg in steps:
                if kind == 0:
                    got = eng.analysis_pyramid(cur, dec_lo, dec_hi, mode_id, arg)
                elif kind == 1:
                    got = eng.analysis_pair(cur, dec_lo, dec_hi, mode_id)
                elif kind == 2:
                    got = eng.analysis_tail(cur, dec_lo, dec_hi, mode_id, arg)
                else:
                    got = (eng.analysis(cur, dec_lo, dec_hi, mode_id),)
                if got is None:  # (cannot happen while the memo is cleared with the plans; never guess: take the long way)
                    break
                bufs.extend(got)
                cur = got[-1][:, 0]
            else:
                bufs.reverse()
                return layout, cur, bufs
            _route_memo.pop(rkey, None)
            bufs, cur = [], x
    steps = []
    while done < level:
        mode_id = _mode_id(mode)
        _check_pad(cur.shape[1:], flen, "reflect" if mode is None else mode)
        differentiable = torch.is_grad_enabled() and (cur.requires_grad or tap_t is not None)
        if ndim == 2 and not on_device and (not differentiable or tap_t is None):
            # several levels per launch (three of a big plane, the whole pyramid of a small one), the approximations between them kept
            # on chip (mifwt_dwt2_fwd_pyramid); the pad
            # checks of the fused trips are the reference's own and run before anything is launched
            want = min(_engine.MAX_PYRAMID_LEVELS, level - done)
            ns = list(cur.shape[1:])
            for _l in range(want):
                _check_pad(ns, flen, "reflect" if mode is None else mode)
                ns = [(n + flen - 1) // 2 for n in ns]
            if differentiable:
                # gradients w.r.t. the data only (a learnable filter bank takes the per-level ops, which also produce tap gradients):
                # the same launch as a differentiable op
                if _engine.ENGINE.pyramid_levels(cur, flen, mode_id, want) > 0:
                    bands = _AnalysisPyramid.apply(cur, dec_lo, dec_hi, mode_id, want)
                    cur = bands[0]
                    nfused = (len(bands) - 1) // 3
                    bufs.extend(bands[1 + 3 * l : 4 + 3 * l] for l in range(nfused))  # (a level as its three bands: pack_2d / pack_dict take either form)
                    done += nfused
                    continue
            else:
                pyr = _engine.ENGINE.analysis_pyramid(cur, dec_lo, dec_hi, mode_id, want)
                if pyr is not None:
                    steps.append((0, want))
                    bufs.extend(pyr)
                    cur = pyr[-1][:, 0]
                    done += len(pyr)
                    continue
        if ndim == 2 and level - done >= 2 and not differentiable and not on_device:
            # two levels per launch, the approximation between them kept on chip (mifwt_dwt2_fwd_pair); the second
            # level's reflect / periodic pad check is the reference's own (it would raise inside the next trip)
            n1 = [(n + flen - 1) // 2 for n in cur.shape[1:]]
            _check_pad(n1, flen, "reflect" if mode is None else mode)
            pair = _engine.ENGINE.analysis_pair(cur, dec_lo, dec_hi, mode_id)
            if pair is not None:
                steps.append((1, 0))
                bufs.extend(pair)
                cur = pair[1][:, 0]
                done += 2
                continue
        if ndim == 1 and level - done >= 2 and not on_device and (not differentiable or tap_t is None):
            # the deep levels of a 1-D pyramid in one launch (mifwt_dwt1_fwd_tail) once a row fits into LDS, several levels of
            # longer rows per launch before that (mifwt_dwt1_fwd_long); the pad checks of the fused trips are the reference's
            # own and run before anything is launched
            n = cur.shape[1]
            for _l in range(level - done):
                _check_pad([n], flen, "reflect" if mode is None else mode)
                n = (n + flen - 1) // 2
            if differentiable:  # (gradients w.r.t. the data only: the same launches as a differentiable op)
                try:
                    bands = _AnalysisTail.apply(cur, dec_lo, dec_hi, mode_id, level - done)
                except _FusedRouteUnavailable:
                    bands = None
                if bands is not None:
                    cur = bands[0]
                    bufs.extend((d,) for d in bands[1:])  # (a level as its detail row: pack_1d takes either form)
                    done += len(bands) - 1
                    continue
            else:
                tail = _engine.ENGINE.analysis_tail(cur, dec_lo, dec_hi, mode_id, level - done)
                if tail is not None:
                    steps.append((2, level - done))
                    bufs.extend(tail)
                    cur = tail[-1][:, 0]
                    done += len(tail)
                    continue
        done += 1
        if differentiable:
            buf = _AnalysisLevel.apply(cur, dec_lo, dec_hi, mode_id, *((tap_t[0], tap_t[1]) if tap_t else (None, None)))
        else:
            steps.append((3, 0))
            buf = _engine.ENGINE.analysis(cur, dec_lo, dec_hi, mode_id)
        bufs.append(buf)
        cur = buf[:, 0]
    if rkey is not None:
        if len(_route_memo) > 512:
            _route_memo.clear()
        _route_memo[rkey] = tuple(steps)
    bufs.reverse()
    return layout, cur, bufs


# ------------------------------------------------------------------------------------------ synthesis
# (both memos hold idempotent values and are only touched through single dict operations — get / item assignment / clear — each atomic
# under the GIL; a thread that loses a race recomputes the same entry)
_tail_memo: dict = {}  # geometry of a 2-D reconstruction -> (levels the streaming launch takes, final extents, its plan)
_small_memo: dict = {}  # ... of a small plane -> (final extents, plan of the one-launch reconstruction)
_engine._routing_caches.append(_tail_memo)
_engine._routing_caches.append(_small_memo)


def _adjust_trim(res_size: int, next_size: int) -> int:
    """src/ptwt/_util.py:231-244 on the already L-2-cropped size."""
    if next_size == res_size:
        return 0
    if next_size == res_size - 1:
        return 1
    raise AssertionError("padding error, please check if dec and rec wavelets are identical.")


def _check_same_device_dtype(tensors: Sequence[torch.Tensor]) -> None:
    """src/ptwt/_util.py:307-348."""
    first = tensors[0]
    for t in tensors:
        if t.device != first.device:
            raise ValueError("coefficients must be on the same device")
    for t in tensors:
        if t.dtype != first.dtype:
            raise ValueError("coefficients must have the same dtype")


def synthesis(approx: torch.Tensor, levels: List[List[torch.Tensor]], wavelet, axes: AxisHint, ndim: int,
              separable: bool) -> torch.Tensor:
    """Multi-level synthesis.  ``levels``: per level (coarsest first) the 2^n-1 detail tensors in band order."""
    axes = _ensure_axes(axes, ndim)
    layout = _Layout(approx, ndim, axes)
    flat = [approx] + [t for lvl in levels for t in lvl]
    for t in flat:
        if not isinstance(t, torch.Tensor):
            raise ValueError(f"Unexpected input type {type(t)}")
    _check_same_device_dtype(flat)
    dbank = device_bank(wavelet, approx.device) if approx.is_cuda else None
    if dbank is not None:  # (see analysis)
        rec_lo, rec_hi = _engine.DevTaps(dbank[2]), _engine.DevTaps(dbank[3])
    else:
        _, _, rec_lo, rec_hi = host_taps(wavelet)
    on_device = dbank is not None
    tap_t = _tap_tensors(wavelet)
    flen = len(rec_lo)
    cur = layout.fold(approx)
    folded = [[layout.fold(t) for t in lvl] for lvl in levels]
    def level_out_extent(cur_shape, pos):
        """Checks of one trip of the reference's level loop (shapes, trims) -> output extents of that level."""
        det = folded[pos]
        if separable:
            # the separable reference crops the running approximation to the detail shape
            # (src/ptwt/separable_conv_transform.py:94-97) and never trims the synthesis output
            cur_shape = tuple(min(c, s_) for c, s_ in zip(cur_shape, det[0].shape))
            trims = [0] * ndim
        else:
            trims = [0] * ndim
            if pos + 1 < len(folded):
                nxt = folded[pos + 1][0].shape
                trims = [_adjust_trim(2 * cur_shape[1 + a] - flen + 2, nxt[1 + a]) for a in range(ndim)]
        for t in det:
            if tuple(t.shape) != tuple(cur_shape):
                if ndim == 1:  # torch.stack in the reference (src/ptwt/conv_transform.py:186)
                    raise RuntimeError("stack expects each tensor to be equal size")
                raise ValueError("All coefficients on each level must have the same shape")
        out_ext = [2 * cur_shape[1 + a] - flen + 2 - trims[a] for a in range(ndim)]
        if min(out_ext) < 1:
            raise ValueError("coefficients too short for this wavelet")
        return out_ext

    pos = 0
    if ndim == 1 and not separable and len(folded) >= 2 and not on_device:
        # the coarse levels of a 1-D reconstruction in one launch (mifwt_dwt1_inv_tail) while a level's output still fits into
        # LDS; every fused trip passes the reference's own checks first
        differentiable = torch.is_grad_enabled() and (tap_t is not None or any(t.requires_grad for t in flat))
        if not differentiable or tap_t is None:
            try:
                outs, shape = [], tuple(cur.shape)
                for lv in range(len(folded)):
                    ext = level_out_extent(shape, lv)
                    outs.append(ext[0])
                    shape = (shape[0], ext[0])
            except (ValueError, RuntimeError, AssertionError):
                outs = []  # the per-level loop below raises the reference's error at the level it belongs to
            cap = 16384 if cur.dtype == torch.float32 else 8192
            dets = [lvl[0] for lvl in folded]
            eng = _engine.ENGINE

            def fuse(fuse, cur, a, b):  # (itself as an argument: a closure over its own name would be a reference cycle per call)
                """Levels a .. b-1 with as few launches as possible: the finest ones in the chunked launch (mifwt_dwt1_inv_long:
                as many as its halo rule allows), what is coarser first — chunked as well while there are too few rows for one
                workgroup each, else in the one-workgroup-per-row launch (mifwt_dwt1_inv_tail) while the outputs fit into LDS."""
                if b - a >= 2:
                    y, k = eng.synthesis_long(cur, dets[a:b], rec_lo, rec_hi, outs[a:b])
                    if y is not None:
                        return y
                    if 2 <= k < b - a:
                        cur = fuse(fuse, cur, a, b - k)
                        a = b - k
                        y, _k = eng.synthesis_long(cur, dets[a:b], rec_lo, rec_hi, outs[a:b])
                        if y is not None:
                            return y
                    nf = 0
                    while a + nf < b and outs[a + nf] <= cap:
                        nf += 1
                    if nf >= 2:
                        y = eng.synthesis_tail(cur, dets[a:a + nf], rec_lo, rec_hi, outs[a:a + nf])
                        if y is not None:
                            cur, a = y, a + nf
                while a < b:
                    cur = eng.synthesis(cur, folded[a], rec_lo, rec_hi, [outs[a]])
                    a += 1
                return cur

            if len(outs) == len(folded):
                if differentiable:  # (gradients w.r.t. the coefficients only: the same launches as ONE differentiable op)
                    cur = _SynthesisChain1d.apply(lambda a0, dd: fuse(fuse, a0, 0, len(dd)), rec_lo, rec_hi, cur, *dets)  # (fuse reads `dets`: the same tensors)
                else:
                    cur = fuse(fuse, cur, 0, len(folded))
                pos = len(folded)
    any_grad = torch.is_grad_enabled() and (tap_t is not None or any(t.requires_grad for t in flat))
    # (gradients w.r.t. the coefficients only: the multi-level launches as differentiable ops, _SynthesisPyramid; a learnable filter
    # bank takes the per-level ops.  The separable containers' crops of the running approximation take its LEADING samples, like the
    # reference's trims: the adjoint of a level sees a gradient of the cropped extents and treats the rest as zeros — same backward)
    fused_grad = any_grad and tap_t is None
    gkey = None
    if ndim == 2 and folded and not on_device and (not any_grad or fused_grad):
        # geometry of the call (every band's shape: the reference's shape checks are part of what is remembered)
        gkey = (cur.dtype, cur.shape, cur.stride(), tuple((lv[0].stride(), *[t.shape for t in lv]) for lv in folded), flen, separable)
    # (the finest level's four coefficient planes alone must fit into LDS: 10 240 samples each at most)
    if gkey is not None and folded[-1][0].shape[-1] * folded[-1][0].shape[-2] <= 10240:
        # every level of a small plane in one launch, the running approximation kept on chip (mifwt_dwt2_inv_pyramid); every
        # fused trip passes the reference's own checks first (remembered per geometry)
        hit = _small_memo.get(gkey)
        if hit is None:
            try:
                shape, out_ext = tuple(cur.shape), None
                for lv in range(len(folded)):
                    out_ext = level_out_extent(shape, lv)
                    shape = (shape[0], *out_ext)
            except (ValueError, RuntimeError, AssertionError):
                out_ext = None  # the per-level loop below raises the reference's error at the level it belongs to
            pl = _engine.ENGINE.synthesis_pyramid_plan(cur, folded, flen, out_ext) if out_ext is not None and cur.dtype == torch.float32 else None
            if len(_small_memo) > 1024:
                _small_memo.clear()
            hit = _small_memo[gkey] = (out_ext, pl)
        out_ext, pl = hit
        if pl is not None and pl[3]:
            if fused_grad:
                y = _SynthesisPyramid.apply(rec_lo, rec_hi, tuple(out_ext), pl, len(folded), cur, *[t for lv in folded for t in lv])
            else:
                y = _engine.ENGINE.synthesis_pyramid(cur, folded, rec_lo, rec_hi, out_ext, plan=pl)
            if y is not None:
                return layout.unfold(y)
    # big planes: the FINEST up to three levels (that is where the bytes are) in one streaming launch (mifwt_dwt2_inv_pyramid's
    # second kernel); what is coarser goes first, through the loop below.  `tail` = how many levels that launch takes (0: none).
    # The decision depends on the geometry only and is remembered per geometry.
    tail, tail_ext, tail_plan = 0, None, None
    if gkey is not None and cur.dtype == torch.float32 and folded[-1][0].shape[-1] * folded[-1][0].shape[-2] > 10240:
        tkey = gkey
        hit = _tail_memo.get(tkey)
        if hit is None:
            hit = (0, None, None)
            try:
                shapes, shape = [], tuple(cur.shape)
                for lv in range(len(folded)):
                    shapes.append(shape)
                    shape = (shape[0], *level_out_extent(shape, lv))
                final_ext = shape[1:]
            except (ValueError, RuntimeError, AssertionError):
                shapes = None  # the per-level loop below raises the reference's error at the level it belongs to
            if shapes is not None:
                for k in (3, 2, 1):
                    if k > len(folded):
                        continue
                    first = len(folded) - k
                    if first == 0:
                        a0 = cur
                    else:  # the approximation the coarser levels will hand over: dense, cropped to the level's band extents
                        ext = tuple(min(c, s_) for c, s_ in zip(shapes[first], folded[first][0].shape)) if separable else shapes[first]
                        a0 = torch.empty(ext, dtype=cur.dtype, device="meta")
                    if separable and first == 0:
                        a0 = a0[tuple(slice(0, s_) for s_ in folded[0][0].shape)]
                    pl = _engine.ENGINE.synthesis_pyramid_plan(a0, folded[first:], flen, final_ext)
                    if pl is not None and pl[3] == 2:
                        hit = (k, final_ext, pl)
                        break
            if len(_tail_memo) > 1024:
                _tail_memo.clear()
            _tail_memo[tkey] = hit
        tail, tail_ext, tail_plan = hit
    while pos < len(folded):
        det = folded[pos]
        if separable:
            cur = cur[tuple(slice(0, s_) for s_ in det[0].shape)]
        if tail and pos == len(folded) - tail:
            # (the plan was made for a dense hand-over approximation; a strided one — a separable crop — is looked up afresh)
            same = pos == 0 or cur.is_contiguous()
            if fused_grad:
                y = _SynthesisPyramid.apply(rec_lo, rec_hi, tuple(tail_ext), tail_plan if same else None, len(folded) - pos, cur,
                                            *[t for lv in folded[pos:] for t in lv])
            else:
                y = _engine.ENGINE.synthesis_pyramid(cur, folded[pos:], rec_lo, rec_hi, tail_ext, plan=tail_plan if same else None)
            if y is not None:
                return layout.unfold(y)
            tail = 0  # (bands that do not share their strides: level by level)
        out_ext = level_out_extent(tuple(cur.shape), pos)
        differentiable = torch.is_grad_enabled() and (cur.requires_grad or any(t.requires_grad for t in det) or tap_t is not None)
        # levels go in pairs counted from the FINEST one of those left to this loop (that is where the bytes are): an odd count
        # starts with a single level
        if (ndim == 2 and not differentiable and not on_device and (len(folded) - tail - pos) % 2 == 0 and len(folded) - tail - pos >= 2
                and not (torch.is_grad_enabled() and any(t.requires_grad for t in folded[pos + 1]))):
            # two levels per launch, the approximation between them kept on chip (mifwt_dwt2_inv_pair); the checks of the
            # second trip are the reference's own and run before anything is launched.  Separable containers: the crop of the
            # running approximation to the next detail shape IS the extent the kernel synthesises the approximation tile for
            out_ext2 = level_out_extent((cur.shape[0], *out_ext), pos + 1)
            y = _engine.ENGINE.synthesis_pair(cur, det, folded[pos + 1], rec_lo, rec_hi, out_ext2)
            if y is not None:
                cur = y
                pos += 2
                continue
        if differentiable:
            cur = _SynthesisLevel.apply(rec_lo, rec_hi, tuple(out_ext), *((tap_t[2], tap_t[3]) if tap_t else (None, None)), cur, *det)
        else:
            cur = _engine.ENGINE.synthesis(cur, det, rec_lo, rec_hi, out_ext)
        pos += 1
    return layout.unfold(cur)


# ------------------------------------------------------------------------------------------ containers
def pack_1d(layout: _Layout, approx, bufs) -> List[torch.Tensor]:
    # (the detail row is the last plane of a level buffer — [B, 2, M] or [B, 1, M] — or the one band of a differentiable multi-level launch)
    return [layout.unfold(approx)] + [layout.unfold(b[-1] if isinstance(b, tuple) else b[:, -1]) for b in bufs]


def pack_2d(layout: _Layout, approx, bufs):
    out = [layout.unfold(approx)]
    unfold = layout.unfold
    for b in bufs:  # (H, V, D) = ('da', 'ad', 'dd') = bands 2, 1, 3 (one unbind: a third of the host time of three b[:, k])
        # ([B, 4, ..] with the approximation in plane 0, [B, 3, ..] from a multi-level launch, or the three bands of a differentiable one)
        ad, da, dd = b if isinstance(b, tuple) else b.unbind(1)[-3:]
        out.append(WaveletDetailTuple2d(unfold(da), unfold(ad), unfold(dd)))
    return tuple(out)


def pack_dict(layout: _Layout, approx, bufs, keys: Sequence[str]):
    out: list = [layout.unfold(approx)]
    unfold = layout.unfold
    keys = tuple(keys)
    idx = _BAND_OF_KEYS.get(keys)
    if idx is None:
        idx = _BAND_OF_KEYS[keys] = tuple(_band(k) for k in keys)  # (the string arithmetic of _band per key and call was 8 us of a wavedec3)
    nb = 1 << len(keys[0])
    for b in bufs:
        planes = b if isinstance(b, tuple) else b.unbind(1)
        off = nb - len(planes)  # 1 for a details-only buffer of a multi-level launch (plane s - 1 = band s)
        out.append({k: unfold(planes[i - off]) for k, i in zip(keys, idx)})
    return tuple(out)


_BAND_OF_KEYS: dict = {}


def unpack_dict_levels(coeffs, ndim: int, what: str) -> List[List[torch.Tensor]]:
    """dict levels -> band-ordered lists; validates like src/ptwt/conv_transform_3.py:194-204 /
    separable_conv_transform.py:174-177."""
    nb = 1 << ndim
    levels = []
    for c in coeffs[1:]:
        if not isinstance(c, dict) or len([k for k in c if k != "a" * ndim]) != nb - 1:
            raise ValueError(
                f"Unexpected detail coefficient type: {type(c)}. Detail coefficients must be a dict containing "
                f"{nb - 1} tensors as returned by {what}."
            )
        try:
            levels.append([c[format(s, f"0{ndim}b").replace("0", "a").replace("1", "d")] for s in range(1, nb)])
        except KeyError as e:
            raise ValueError(f"missing detail coefficient key {e} for a {ndim}D transform") from None
    return levels
