"""ctypes binding of ``libmifwt.so`` (C ABI: include/mifwt.h) — the only compute backend of this package.

There is deliberately NO CPU or eager-PyTorch fallback here: tensors must live on a ROCm device and the
HIP library must have been built (``python -c "import __graft_entry__ as g; g.build()"``); anything else
raises.  PyTorch is used for device memory (caching allocator) and streams only.
"""
from __future__ import annotations

import ctypes
import os
from typing import List, Optional, Sequence, Tuple

import torch

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, "libmifwt.so")
ABI_VERSION = 1

MODE_IDS = {"zero": 0, "constant": 1, "reflect": 2, "periodic": 3, "symmetric": 4}
_DTYPE_IDS = {torch.float32: 0, torch.float64: 1, torch.float16: 2}

_i64x3 = ctypes.c_int64 * 3
_i64x4 = ctypes.c_int64 * 4


class LevelDesc(ctypes.Structure):
    """Mirror of ``mifwt_level_desc`` (include/mifwt.h)."""

    _fields_ = [
        ("ndim", ctypes.c_int32),
        ("dtype", ctypes.c_int32),
        ("mode", ctypes.c_int32),
        ("filt_len", ctypes.c_int32),
        ("batch", ctypes.c_int64),
        ("sig_extent", _i64x3),
        ("sig_stride", _i64x4),
        ("coef_extent", _i64x3),
        ("approx_stride", _i64x4),
        ("detail_stride", _i64x4),
    ]


_lib: Optional[ctypes.CDLL] = None


def load_library() -> ctypes.CDLL:
    """Load libmifwt.so (once).  Fails loudly when the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(
            f"ptwt_amd: HIP extension {LIB_PATH} is missing — build it with "
            "`python -c \"import __graft_entry__ as g; g.build()\"` (hipcc --offload-arch=gfx950). "
            "There is no CPU/eager fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    vp, cp = ctypes.c_void_p, ctypes.c_char_p
    dbl_p = ctypes.POINTER(ctypes.c_double)
    desc_p = ctypes.POINTER(LevelDesc)
    lib.mifwt_abi_version.restype = ctypes.c_int
    lib.mifwt_abi_version.argtypes = []
    lib.mifwt_strerror.restype = cp
    lib.mifwt_strerror.argtypes = [ctypes.c_int]
    lib.mifwt_kernel_id.restype = ctypes.c_int
    lib.mifwt_kernel_id.argtypes = [desc_p, ctypes.c_int]
    lib.mifwt_workspace_bytes.restype = ctypes.c_size_t
    lib.mifwt_workspace_bytes.argtypes = [desc_p, ctypes.c_int]
    lib.mifwt_dwt_fwd.restype = ctypes.c_int
    lib.mifwt_dwt_fwd.argtypes = [desc_p, vp, vp, ctypes.POINTER(vp), dbl_p, dbl_p, vp, ctypes.c_size_t, vp]
    lib.mifwt_dwt_inv.restype = ctypes.c_int
    lib.mifwt_dwt_inv.argtypes = [desc_p, vp, ctypes.POINTER(vp), vp, dbl_p, dbl_p, vp, ctypes.c_size_t, vp]
    lib.mifwt_set_option.restype = ctypes.c_int
    lib.mifwt_set_option.argtypes = [ctypes.c_int, ctypes.c_int]
    if lib.mifwt_abi_version() != ABI_VERSION:
        raise RuntimeError("ptwt_amd: libmifwt.so ABI version mismatch; rebuild the extension")
    _lib = lib
    return lib


def _check(rc: int) -> None:
    if rc != 0:
        msg = load_library().mifwt_strerror(rc).decode()
        raise RuntimeError(f"libmifwt: {msg} (code {rc})")


def _taps_array(taps: Sequence[float]):
    return (ctypes.c_double * len(taps))(*taps)


def _require_gpu(t: torch.Tensor) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"ptwt_amd: expected a tensor on a ROCm device, got device '{t.device}'. This engine runs on "
            "MI355X only; there is no CPU path."
        )


# Optional per-level device timing (bench.py's roofline leg): when set to a list, every level call
# appends (tag, kernel_id, start_event, end_event) recorded on the launch stream.
level_events: Optional[list] = None

OPT_FORCE_GENERIC = 0
OPT_ROWS_PER_CHUNK = 1
OPT_PREFETCH_PAIRS = 2
OPT_COOP = 3
OPT_NT_STORE = 4


def set_option(key: int, value: int) -> None:
    """Library-wide test/diagnostic switches (e.g. ``OPT_FORCE_GENERIC`` to bypass the fused kernels)."""
    _check(load_library().mifwt_set_option(key, value))


class HipLevelEngine:
    """One decomposition / reconstruction level for a folded batch, on the GPU, through the C ABI."""

    def analysis(self, x: torch.Tensor, dec_lo: Sequence[float], dec_hi: Sequence[float], mode_id: int) -> torch.Tensor:
        """``x``: [B, N_0..N_{n-1}] (any strides) -> one buffer [B, 2^n, M_0..] whose plane ``s`` is band ``s``
        (bit (n-1-a) of s set <=> high-pass along axis a; plane 0 = approximation)."""
        _require_gpu(x)
        lib = load_library()
        ndim = x.dim() - 1
        flen = len(dec_lo)
        batch = x.shape[0]
        sig = [int(n) for n in x.shape[1:]]
        coef = [(n + 2 * ((2 * flen - 3) // 2) + (n % 2) - flen) // 2 + 1 for n in sig]
        nb = 1 << ndim
        buf = torch.empty((batch, nb, *coef), dtype=x.dtype, device=x.device)
        if buf.numel() == 0:
            return buf
        d = LevelDesc()
        d.ndim, d.dtype, d.mode, d.filt_len, d.batch = ndim, _DTYPE_IDS[x.dtype], mode_id, flen, batch
        bstride = buf.stride()
        for a in range(ndim):
            d.sig_extent[a] = sig[a]
            d.coef_extent[a] = coef[a]
            d.sig_stride[1 + a] = x.stride(1 + a)
            d.approx_stride[1 + a] = d.detail_stride[1 + a] = bstride[2 + a]
        d.sig_stride[0] = x.stride(0)
        d.approx_stride[0] = d.detail_stride[0] = bstride[0]
        base = buf.data_ptr()
        plane = bstride[1] * buf.element_size()
        details = (ctypes.c_void_p * (nb - 1))(*[base + s * plane for s in range(1, nb)])
        self._run(lib, d, 0, x, lambda ws, wsb, stream: lib.mifwt_dwt_fwd(
            ctypes.byref(d), x.data_ptr(), base, details, _taps_array(dec_lo), _taps_array(dec_hi), ws, wsb, stream))
        return buf

    def synthesis(self, approx: torch.Tensor, details: List[torch.Tensor], rec_lo: Sequence[float],
                  rec_hi: Sequence[float], out_extent: Sequence[int]) -> torch.Tensor:
        """``approx`` and the 2^n-1 ``details`` (band order): [B, M_0..] -> y [B, *out_extent] (dense)."""
        _require_gpu(approx)
        lib = load_library()
        ndim = approx.dim() - 1
        flen = len(rec_lo)
        batch = approx.shape[0]
        y = torch.empty((batch, *out_extent), dtype=approx.dtype, device=approx.device)
        if y.numel() == 0:
            return y
        ref_stride = details[0].stride()
        details = [t if t.stride() == ref_stride else t.contiguous() for t in details]
        if any(t.stride() != details[0].stride() for t in details):
            details = [t.contiguous() for t in details]
        d = LevelDesc()
        d.ndim, d.dtype, d.mode, d.filt_len, d.batch = ndim, _DTYPE_IDS[approx.dtype], 0, flen, batch
        for a in range(ndim):
            d.sig_extent[a] = int(out_extent[a])
            d.coef_extent[a] = int(approx.shape[1 + a])
        for a in range(ndim + 1):
            d.sig_stride[a] = y.stride(a)
            d.approx_stride[a] = approx.stride(a)
            d.detail_stride[a] = details[0].stride(a)
        dptr = (ctypes.c_void_p * len(details))(*[t.data_ptr() for t in details])
        self._run(lib, d, 1, approx, lambda ws, wsb, stream: lib.mifwt_dwt_inv(
            ctypes.byref(d), approx.data_ptr(), dptr, y.data_ptr(), _taps_array(rec_lo), _taps_array(rec_hi), ws, wsb, stream))
        return y

    @staticmethod
    def _run(lib, d: LevelDesc, direction: int, anchor: torch.Tensor, call) -> None:
        dev = anchor.device
        if dev.index is not None and dev.index != torch.cuda.current_device():
            with torch.cuda.device(dev):
                return HipLevelEngine._run(lib, d, direction, anchor, call)
        wsb = lib.mifwt_workspace_bytes(ctypes.byref(d), direction)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev) if wsb else None
        stream = torch.cuda.current_stream(dev)
        ev = None
        if level_events is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record(stream)
        rc = call(ws.data_ptr() if ws is not None else None, wsb, stream.cuda_stream)
        if ev is not None:
            ev[1].record(stream)
            kid = lib.mifwt_kernel_id(ctypes.byref(d), direction)
            level_events.append((("inv" if direction else "fwd"), kid, tuple(d.sig_extent[: d.ndim]), ev[0], ev[1]))
        _check(rc)
        if ws is not None:
            ws.record_stream(stream)  # scratch is released to the allocator only after the level has run


def kernel_id(ndim: int, dtype: torch.dtype, mode: str, filt_len: int, batch: int, sig_extent: Sequence[int],
              direction: int = 0) -> int:
    """Which kernel family a dense, default-layout level of this geometry dispatches to (0 = generic)."""
    lib = load_library()
    d = LevelDesc()
    d.ndim, d.dtype, d.mode, d.filt_len, d.batch = ndim, _DTYPE_IDS[dtype], MODE_IDS[mode], filt_len, batch
    if direction == 0:
        coef = [(n + filt_len - 1) // 2 for n in sig_extent]
        sig = list(sig_extent)
    else:
        sig = list(sig_extent)
        coef = [(n + filt_len - 1) // 2 for n in sig_extent]
    st = 1
    for a in reversed(range(ndim)):
        d.sig_extent[a], d.coef_extent[a] = sig[a], coef[a]
    s = 1
    for a in reversed(range(ndim)):
        d.sig_stride[1 + a] = s
        s *= sig[a]
    d.sig_stride[0] = s
    s = 1
    for a in reversed(range(ndim)):
        d.approx_stride[1 + a] = d.detail_stride[1 + a] = s
        s *= coef[a]
    d.approx_stride[0] = d.detail_stride[0] = s * (1 << ndim)
    return lib.mifwt_kernel_id(ctypes.byref(d), direction)


ENGINE = HipLevelEngine()
