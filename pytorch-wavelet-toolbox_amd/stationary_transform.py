"""Stationary (undecimated) wavelet transform: ``swt`` / ``iswt`` (API of reference src/ptwt/stationary_transform.py).

Equivalent to ``pywt.swt(..., trim_approx=True, norm=False)`` like the reference.  Each level is one HIP kernel
(C ABI ``mifwt_swt_fwd`` / ``mifwt_swt_inv``): stride-1 filter bank with dilation ``2^level`` and the periodic
extension as an index map — the reference's ``_circular_pad`` + ``F.conv1d(dilation)`` + ``split`` (:95-107) and
``stack`` + ``_circular_pad`` + grouped ``F.conv_transpose1d`` + ``mean`` (:142-156).  Differentiable w.r.t. the data
(each level kernel is the other's adjoint with reversed taps).
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence, Union

import torch

from . import _engine, _fwt
from ._wavelets import host_taps
from .constants import Wavelet, supported_dtypes

__all__ = ["swt", "iswt"]

_bound = False


def _lib():
    global _bound
    lib = _engine.load_library()
    if not _bound:
        vp, i64, dbl_p = ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_double)
        lib.mifwt_swt_fwd.restype = ctypes.c_int
        lib.mifwt_swt_fwd.argtypes = [ctypes.c_int, ctypes.c_int, i64, i64, i64, vp, i64, vp, vp, i64, i64, dbl_p, dbl_p,
                                      ctypes.c_double, vp]
        lib.mifwt_swt_inv.restype = ctypes.c_int
        lib.mifwt_swt_inv.argtypes = [ctypes.c_int, ctypes.c_int, i64, i64, i64, vp, vp, i64, i64, vp, i64, dbl_p, dbl_p,
                                      ctypes.c_double, vp]
        _bound = True
    return lib


def _stream(t: torch.Tensor) -> int:
    return _engine._raw_stream(t.device.index if t.device.index is not None else torch.cuda.current_device())


def _rows(t: torch.Tensor) -> torch.Tensor:
    """[B, N] with contiguous samples (row stride free)."""
    return t if t.stride(-1) == 1 else t.contiguous()


def _level_fwd(x: torch.Tensor, lo: Sequence[float], hi: Sequence[float], dilation: int, scale: float) -> torch.Tensor:
    """x [B, N] -> buffer [B, 2, N]: plane 0 low-pass, plane 1 high-pass."""
    _engine._require_gpu(x)
    x = _rows(x)
    b, n = x.shape
    buf = torch.empty((b, 2, n), dtype=x.dtype, device=x.device)
    if buf.numel() == 0:
        return buf
    with torch.cuda.device(x.device):
        rc = _lib().mifwt_swt_fwd(_engine._DTYPE_IDS[x.dtype], len(lo), b, n, dilation, x.data_ptr(), x.stride(0),
                                  buf.data_ptr(), buf.data_ptr() + n * buf.element_size(), 2 * n, 2 * n,
                                  _engine._taps_array(lo), _engine._taps_array(hi), scale, _stream(x))
    _engine._check(rc)
    return buf


def _level_inv(a: torch.Tensor, d: torch.Tensor, lo: Sequence[float], hi: Sequence[float], dilation: int,
               scale: float) -> torch.Tensor:
    _engine._require_gpu(a)
    a, d = _rows(a), _rows(d)
    b, n = a.shape
    y = torch.empty((b, n), dtype=a.dtype, device=a.device)
    if y.numel() == 0:
        return y
    with torch.cuda.device(a.device):
        rc = _lib().mifwt_swt_inv(_engine._DTYPE_IDS[a.dtype], len(lo), b, n, dilation, a.data_ptr(), d.data_ptr(),
                                  a.stride(0), d.stride(0), y.data_ptr(), n, _engine._taps_array(lo),
                                  _engine._taps_array(hi), scale, _stream(a))
    _engine._check(rc)
    return y


# ---- learnable filter banks, second order (the per-level maps closed under differentiation, as _fwt._Axis1 … for the decimated levels) ----
# A stationary level is bilinear in (signal, taps):  lo[n] = s sum_m h[m] x[(n + D (L/2 - m)) mod N].  Analysis A(h) x, its transpose
# A(h)^T g (= the synthesis kernel with reversed taps) and the tap correlation C(x, g) form a set closed under differentiation; the
# synthesis y = S(r)(a, d), S(r)^T g_y (= the analysis kernel with reversed taps) and C'(a, d, g_y) likewise.
def _rev(t: torch.Tensor) -> torch.Tensor:
    return t.flip(0)


class _Swt1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, lo_t, hi_t, dilation, scale):
        ctx.meta = (dilation, scale)
        ctx.save_for_backward(x, lo_t, hi_t)
        return _level_fwd(x, _fwt._host_floats_of(lo_t), _fwt._host_floats_of(hi_t), dilation, scale)

    @staticmethod
    def backward(ctx, g):
        x, lo_t, hi_t = ctx.saved_tensors
        dilation, scale = ctx.meta
        g_x = _Iswt1.apply(g[:, 0], g[:, 1], _rev(lo_t), _rev(hi_t), dilation, scale) if ctx.needs_input_grad[0] else None
        t_lo = t_hi = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            t_lo, t_hi = _Swt1Corr.apply(x, g, lo_t.numel(), dilation, scale)
            t_lo, t_hi = _fwt._like(t_lo, lo_t), _fwt._like(t_hi, hi_t)
        return g_x, t_lo, t_hi, None, None


class _Iswt1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, d, lo_t, hi_t, dilation, scale):
        ctx.meta = (dilation, scale)
        ctx.save_for_backward(a, d, lo_t, hi_t)
        return _level_inv(a, d, _fwt._host_floats_of(lo_t), _fwt._host_floats_of(hi_t), dilation, scale)

    @staticmethod
    def backward(ctx, g_y):
        a, d, lo_t, hi_t = ctx.saved_tensors
        dilation, scale = ctx.meta
        g_a = g_d = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            gb = _Swt1.apply(g_y, _rev(lo_t), _rev(hi_t), dilation, scale)
            g_a, g_d = gb[:, 0], gb[:, 1]
        t_lo = t_hi = None
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
            t_lo, t_hi = _Iswt1Corr.apply(a, d, g_y, lo_t.numel(), dilation, scale)
            t_lo, t_hi = _fwt._like(t_lo, lo_t), _fwt._like(t_hi, hi_t)
        return g_a, g_d, t_lo, t_hi, None, None


class _Swt1Corr(torch.autograd.Function):
    """(x [B, N], g [B, 2, N]) -> (t_lo, t_hi):  t_b[m] = s sum_n g_b[n] x[(n + D L/2 - D m) mod N]."""

    @staticmethod
    def forward(ctx, x, g, flen, dilation, scale):
        ctx.meta = (dilation, scale)
        ctx.save_for_backward(x, g)
        t_lo = torch.zeros(flen, dtype=torch.float64, device=x.device)
        t_hi = torch.zeros_like(t_lo)
        _engine.ENGINE.tap_correlate_dilated(g[:, 0], x, flen, dilation * (flen // 2), -dilation, t_lo)
        _engine.ENGINE.tap_correlate_dilated(g[:, 1], x, flen, dilation * (flen // 2), -dilation, t_hi)
        return t_lo * scale, t_hi * scale

    @staticmethod
    def backward(ctx, w_lo, w_hi):
        x, g = ctx.saved_tensors
        dilation, scale = ctx.meta
        w_lo, w_hi = w_lo.to(x.dtype), w_hi.to(x.dtype)
        g_x = _Iswt1.apply(g[:, 0], g[:, 1], _rev(w_lo), _rev(w_hi), dilation, scale) if ctx.needs_input_grad[0] else None
        g_g = _Swt1.apply(x, w_lo, w_hi, dilation, scale) if ctx.needs_input_grad[1] else None
        return g_x, g_g, None, None, None


class _Iswt1Corr(torch.autograd.Function):
    """(a, d, g_y [B, N]) -> (t_lo, t_hi):  t_lo[j] = s sum_n g_y[n] a[(n + D (L/2 - 1) - D j) mod N]."""

    @staticmethod
    def forward(ctx, a, d, g_y, flen, dilation, scale):
        ctx.meta = (dilation, scale)
        ctx.save_for_backward(a, d, g_y)
        t_lo = torch.zeros(flen, dtype=torch.float64, device=a.device)
        t_hi = torch.zeros_like(t_lo)
        _engine.ENGINE.tap_correlate_dilated(g_y, a, flen, dilation * (flen // 2 - 1), -dilation, t_lo)
        _engine.ENGINE.tap_correlate_dilated(g_y, d, flen, dilation * (flen // 2 - 1), -dilation, t_hi)
        return t_lo * scale, t_hi * scale

    @staticmethod
    def backward(ctx, w_lo, w_hi):
        a, d, g_y = ctx.saved_tensors
        dilation, scale = ctx.meta
        w_lo, w_hi = w_lo.to(a.dtype), w_hi.to(a.dtype)
        g_a = g_d = g_gy = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            gb = _Swt1.apply(g_y, _rev(w_lo), _rev(w_hi), dilation, scale)
            g_a, g_d = gb[:, 0], gb[:, 1]
        if ctx.needs_input_grad[2]:
            g_gy = _Iswt1.apply(a, d, w_lo, w_hi, dilation, scale)
        return g_a, g_d, g_gy, None, None, None


class _SwtLevelGrad(torch.autograd.Function):
    """First-order gradients of a stationary analysis level as an op whose backward has the mixed second derivatives with a learnable
    filter bank (see _fwt._AnalysisLevelGrad)."""

    @staticmethod
    def forward(ctx, g_buf, x, lo_t, hi_t, lo, hi, dilation, scale):
        ctx.meta = (dilation, scale)
        ctx.save_for_backward(g_buf, x, lo_t, hi_t)
        g_x = _level_inv(g_buf[:, 0], g_buf[:, 1], lo[::-1], hi[::-1], dilation, scale)
        L = len(lo)
        t_lo = torch.zeros(L, dtype=torch.float64, device=x.device)
        t_hi = torch.zeros_like(t_lo)
        _engine.ENGINE.tap_correlate_dilated(g_buf[:, 0], x, L, dilation * (L // 2), -dilation, t_lo)
        _engine.ENGINE.tap_correlate_dilated(g_buf[:, 1], x, L, dilation * (L // 2), -dilation, t_hi)
        return g_x, _fwt._like(t_lo * scale, lo_t), _fwt._like(t_hi * scale, hi_t)

    @staticmethod
    def backward(ctx, w_x, w_lo, w_hi):
        g_buf, x, lo_t, hi_t = ctx.saved_tensors
        _fwt._third_order_refused((lo_t, hi_t))
        dilation, scale = ctx.meta
        d = _fwt._partials_at([g_buf, x, lo_t, hi_t], lambda lv: _Swt1.apply(lv[1], lv[2], lv[3], dilation, scale), 0, [w_x, w_lo, w_hi])
        return d[0], d[1], d[2], d[3], None, None, None, None


class _IswtLevelGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g_y, a, d, lo_t, hi_t, lo, hi, dilation, scale):
        ctx.meta = (dilation, scale)
        ctx.save_for_backward(g_y, a, d, lo_t, hi_t)
        g = _level_fwd(g_y, lo[::-1], hi[::-1], dilation, scale)
        L = len(lo)
        t_lo = torch.zeros(L, dtype=torch.float64, device=g_y.device)
        t_hi = torch.zeros_like(t_lo)
        _engine.ENGINE.tap_correlate_dilated(g_y, a, L, dilation * (L // 2 - 1), -dilation, t_lo)
        _engine.ENGINE.tap_correlate_dilated(g_y, d, L, dilation * (L // 2 - 1), -dilation, t_hi)
        return g[:, 0], g[:, 1], _fwt._like(t_lo * scale, lo_t), _fwt._like(t_hi * scale, hi_t)

    @staticmethod
    def backward(ctx, w_a, w_d, w_lo, w_hi):
        g_y, a, d, lo_t, hi_t = ctx.saved_tensors
        _fwt._third_order_refused((lo_t, hi_t))
        dilation, scale = ctx.meta
        out = _fwt._partials_at([g_y, a, d, lo_t, hi_t], lambda lv: _Iswt1.apply(lv[1], lv[2], lv[3], lv[4], dilation, scale), 0,
                                [w_a, w_d, w_lo, w_hi])
        return out[0], out[1], out[2], out[3], out[4], None, None, None, None


class _SwtLevel(torch.autograd.Function):
    """One stationary analysis level (free output scale), differentiable w.r.t. its input and (optionally) the dec taps.  With
    reversed taps the synthesis kernel is its transpose and vice versa, so each Function's backward is the other Function:
    gradients of any order w.r.t. the data, as the reference has them from _circular_pad + conv1d; the tap gradients (first
    order) are the dilated correlation ``mifwt_tap_correlate_dilated``.  ``lo_t`` / ``hi_t``: the tap TENSORS or None (they only tie
    the op into the graph; the kernels take the host copies)."""

    @staticmethod
    def forward(ctx, x, lo, hi, dilation, scale=1.0, lo_t=None, hi_t=None):
        ctx.meta = (lo, hi, dilation, scale)
        ctx.taps = (lo_t, hi_t)
        need_taps = any(t is not None and t.requires_grad for t in (lo_t, hi_t))
        ctx.save_for_backward(x if need_taps else None)
        return _level_fwd(x, lo, hi, dilation, scale)

    @staticmethod
    def backward(ctx, g_buf):
        lo, hi, dilation, scale = ctx.meta
        (x,) = ctx.saved_tensors
        if x is not None and torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ctx.taps):
            # create_graph=True with a learnable filter bank: the same gradients as an op that carries the mixed second derivatives
            g_x, g_lo, g_hi = _SwtLevelGrad.apply(g_buf, x, ctx.taps[0], ctx.taps[1], lo, hi, dilation, scale)
            need = ctx.needs_input_grad
            return g_x if need[0] else None, None, None, None, None, g_lo if need[5] else None, g_hi if need[6] else None
        g_x = _IswtLevel.apply(g_buf[:, 0], g_buf[:, 1], lo[::-1], hi[::-1], dilation, scale) if ctx.needs_input_grad[0] else None
        g_lo = g_hi = None
        if x is not None and (ctx.needs_input_grad[5] or ctx.needs_input_grad[6]):
            # lo[n] = s sum_m h[m] x[(n + D (L/2 - m)) mod N]  =>  dL/dh[m] = s sum_n g_lo[n] x[(n + D L/2 - D m) mod N]
            L = len(lo)
            g_lo = torch.zeros(L, dtype=torch.float64, device=x.device)
            g_hi = torch.zeros_like(g_lo)
            gb = g_buf.detach()
            _engine.ENGINE.tap_correlate_dilated(gb[:, 0], x, L, dilation * (L // 2), -dilation, g_lo)
            _engine.ENGINE.tap_correlate_dilated(gb[:, 1], x, L, dilation * (L // 2), -dilation, g_hi)
            g_lo, g_hi = _fwt._like(g_lo * scale, ctx.taps[0]), _fwt._like(g_hi * scale, ctx.taps[1])
        return g_x, None, None, None, None, g_lo, g_hi


class _IswtLevel(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, d, lo, hi, dilation, scale=0.5, lo_t=None, hi_t=None):
        ctx.meta = (lo, hi, dilation, scale)
        ctx.taps = (lo_t, hi_t)
        need_taps = any(t is not None and t.requires_grad for t in (lo_t, hi_t))
        if need_taps:
            ctx.save_for_backward(a, d)
        else:
            ctx.save_for_backward()
        return _level_inv(a, d, lo, hi, dilation, scale)

    @staticmethod
    def backward(ctx, g_y):
        lo, hi, dilation, scale = ctx.meta
        if ctx.saved_tensors and torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ctx.taps):
            a, d = ctx.saved_tensors
            g_a, g_d, g_lo, g_hi = _IswtLevelGrad.apply(g_y, a, d, ctx.taps[0], ctx.taps[1], lo, hi, dilation, scale)
            need = ctx.needs_input_grad
            return (g_a if need[0] else None, g_d if need[1] else None, None, None, None, None, g_lo if need[6] else None,
                    g_hi if need[7] else None)
        g_a = g_d = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            g = _SwtLevel.apply(g_y, lo[::-1], hi[::-1], dilation, scale)
            g_a, g_d = g[:, 0], g[:, 1]
        g_lo = g_hi = None
        saved = ctx.saved_tensors
        if saved and (ctx.needs_input_grad[6] or ctx.needs_input_grad[7]):
            # y[n] = s sum_j g_lo[j] a[(n + D (L/2 - 1 - j)) mod N] + g_hi[j] d[...]  =>  dL/dg_lo[j] = s sum_n g_y[n] a[(n + D (L/2 - 1) - D j) mod N]
            L = len(lo)
            g_lo = torch.zeros(L, dtype=torch.float64, device=g_y.device)
            g_hi = torch.zeros_like(g_lo)
            gy = g_y.detach()
            _engine.ENGINE.tap_correlate_dilated(gy, saved[0], L, dilation * (L // 2 - 1), -dilation, g_lo)
            _engine.ENGINE.tap_correlate_dilated(gy, saved[1], L, dilation * (L // 2 - 1), -dilation, g_hi)
            g_lo, g_hi = _fwt._like(g_lo * scale, ctx.taps[0]), _fwt._like(g_hi * scale, ctx.taps[1])
        return g_a, g_d, None, None, None, None, g_lo, g_hi


def swt_max_level(input_len: int) -> int:
    """``pywt.swt_max_level``: how often the length can be halved (src/ptwt/stationary_transform.py:93)."""
    level = 0
    while input_len > 0 and input_len % 2 == 0:
        input_len //= 2
        level += 1
    return level


def swt(data: torch.Tensor, wavelet: Union[Wavelet, str], level: Optional[int] = None, *,
        axis: _fwt.AxisHint = None) -> List[torch.Tensor]:
    """Multi-level 1-D stationary transform; returns ``[cA_n, cD_n, ..., cD_1]``, every entry as long as the input
    (drop-in for ``ptwt.swt``, src/ptwt/stationary_transform.py:56-110)."""
    axes = _fwt._ensure_axes(axis, 1)
    layout = _fwt._Layout(data, 1, axes)
    x = layout.fold(data)
    dec_lo, dec_hi, _, _ = host_taps(wavelet)
    tap_t = _fwt._tap_tensors(wavelet)  # learnable filter bank: the taps stay in the graph (src/ptwt/_util.py:115-132)
    if level is None:
        level = swt_max_level(x.shape[-1])
    out: List[torch.Tensor] = []
    cur = x
    for lvl in range(level):
        if torch.is_grad_enabled() and (cur.requires_grad or tap_t is not None):
            buf = _SwtLevel.apply(cur, dec_lo, dec_hi, 2 ** lvl, 1.0, *((tap_t[0], tap_t[1]) if tap_t else (None, None)))
        else:
            buf = _level_fwd(cur, dec_lo, dec_hi, 2 ** lvl, 1.0)
        out.append(layout.unfold(buf[:, 1]))
        cur = buf[:, 0]
    out.append(layout.unfold(cur))
    out.reverse()
    return out


def iswt(coeffs: Sequence[torch.Tensor], wavelet: Union[Wavelet, str], *, axis: _fwt.AxisHint = None) -> torch.Tensor:
    """Inverse of :func:`swt` (drop-in for ``ptwt.iswt``, src/ptwt/stationary_transform.py:113-160)."""
    coeffs = list(coeffs)
    if not coeffs or not isinstance(coeffs[0], torch.Tensor):
        raise ValueError("First element of coeffs must be the approximation coefficient tensor.")
    axes = _fwt._ensure_axes(axis, 1)
    layout = _fwt._Layout(coeffs[0], 1, axes)
    for t in coeffs:
        if not isinstance(t, torch.Tensor):
            raise ValueError(f"Unexpected input type {type(t)}")
    _fwt._check_same_device_dtype(coeffs)
    _, _, rec_lo, rec_hi = host_taps(wavelet)
    tap_t = _fwt._tap_tensors(wavelet)
    cur = layout.fold(coeffs[0])
    details = [layout.fold(t) for t in coeffs[1:]]
    for pos, det in enumerate(details):
        dilation = 2 ** (len(details) - pos - 1)
        if det.shape != cur.shape:
            raise RuntimeError("stack expects each tensor to be equal size")  # torch.stack in the reference (:146)
        if torch.is_grad_enabled() and (cur.requires_grad or det.requires_grad or tap_t is not None):
            cur = _IswtLevel.apply(cur, det, rec_lo, rec_hi, dilation, 0.5, *((tap_t[2], tap_t[3]) if tap_t else (None, None)))
        else:
            cur = _level_inv(cur, det, rec_lo, rec_hi, dilation, 0.5)
    return layout.unfold(cur)
