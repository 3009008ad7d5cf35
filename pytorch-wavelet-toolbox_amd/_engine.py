"""ctypes binding of ``libmifwt.so`` (C ABI: include/mifwt.h) — the only compute backend of this package.

There is deliberately NO CPU or eager-PyTorch fallback here: tensors must live on a ROCm device and the
HIP library must have been built (``python -c "import __graft_entry__ as g; g.build()"``); anything else
raises.  PyTorch is used for device memory (caching allocator) and streams only.
"""
from __future__ import annotations

import ctypes
import os
import threading
import warnings
from typing import List, Optional, Sequence, Tuple

import torch

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, os.environ.get("MIFWT_LIB", "libmifwt.so"))  # (MIFWT_LIB: an experiment build next to the product library, tools/ only)
if "MIFWT_LIB" in os.environ:
    import warnings

    warnings.warn(f"ptwt_amd: MIFWT_LIB is set — running on the experiment build {LIB_PATH}, not on the product library", RuntimeWarning)
ABI_VERSION = 3

MODE_IDS = {"zero": 0, "constant": 1, "reflect": 2, "periodic": 3, "symmetric": 4}
_DTYPE_IDS = {torch.float32: 0, torch.float64: 1, torch.float16: 2}

_i64x3 = ctypes.c_int64 * 3
_i64x4 = ctypes.c_int64 * 4
_array_types: dict = {}


def _arr(base, n: int):
    """ctypes array TYPE ``base * n``, kept alive: ctypes only holds such types weakly, and a type that dies and is rebuilt on every
    call is a reference cycle per call (type <-> its own dictionaries) that only the garbage collector's passes free."""
    t = _array_types.get((base, n))
    if t is None:
        t = _array_types[(base, n)] = base * n
    return t


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
    # the version check comes BEFORE any other symbol is bound: a stale library then says "rebuild" instead of failing with an
    # AttributeError on the first entry point it lacks
    mismatch = lib.mifwt_abi_version() != ABI_VERSION
    if mismatch:
        # (an experiment build loaded through MIFWT_LIB is held to the same check: its mifwt_level_desc / entry-point signatures must be
        # the ones declared above, or a call corrupts memory instead of failing; MIFWT_ALLOW_ABI_MISMATCH=1 is the explicit way around)
        if os.environ.get("MIFWT_ALLOW_ABI_MISMATCH") != "1":
            raise RuntimeError(f"ptwt_amd: {os.path.basename(LIB_PATH)} has ABI version {lib.mifwt_abi_version()}, this package expects "
                               f"{ABI_VERSION}; rebuild the extension (MIFWT_ALLOW_ABI_MISMATCH=1 loads it anyway, at your own risk)")
        warnings.warn("ptwt_amd: ABI version mismatch accepted through MIFWT_ALLOW_ABI_MISMATCH=1")
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
    lib.mifwt_dwt_fwd_adjoint.restype = ctypes.c_int
    lib.mifwt_dwt_fwd_adjoint.argtypes = [desc_p, vp, ctypes.POINTER(vp), vp, dbl_p, dbl_p, vp, ctypes.c_size_t, vp]
    lib.mifwt_dwt_inv_adjoint.restype = ctypes.c_int
    lib.mifwt_dwt_inv_adjoint.argtypes = [desc_p, vp, vp, ctypes.POINTER(vp), dbl_p, dbl_p, vp, ctypes.c_size_t, vp]
    lib.mifwt_tap_correlate.restype = ctypes.c_int
    lib.mifwt_tap_correlate.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, vp, ctypes.c_int64, vp,
                                        ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]
    lib.mifwt_tap_correlate_dilated.restype = ctypes.c_int
    lib.mifwt_tap_correlate_dilated.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_int64, vp, ctypes.c_int64, vp, ctypes.c_int64,
                                                ctypes.c_int, ctypes.c_int64, ctypes.c_int64, vp, vp]
    lib.mifwt_dwt2_fwd_pair_supported.restype = ctypes.c_int
    lib.mifwt_dwt2_fwd_pair_supported.argtypes = [desc_p, desc_p]
    lib.mifwt_dwt2_fwd_pair.restype = ctypes.c_int
    lib.mifwt_dwt2_fwd_pair.argtypes = [desc_p, desc_p, vp, ctypes.POINTER(vp), vp, ctypes.POINTER(vp), dbl_p, dbl_p, vp]
    vpp = ctypes.POINTER(vp)
    lib.mifwt_dwt2_fwd_pyramid_supported.restype = ctypes.c_int
    lib.mifwt_dwt2_fwd_pyramid_supported.argtypes = [ctypes.c_int, ctypes.POINTER(desc_p)]
    lib.mifwt_dwt2_inv_pyramid_supported.restype = ctypes.c_int
    lib.mifwt_dwt2_inv_pyramid_supported.argtypes = [ctypes.c_int, ctypes.POINTER(desc_p)]
    lib.mifwt_dwt2_inv_pyramid.restype = ctypes.c_int
    lib.mifwt_dwt2_inv_pyramid.argtypes = [ctypes.c_int, ctypes.POINTER(desc_p), vp, ctypes.POINTER(vpp), vp, dbl_p, dbl_p, vp]
    lib.mifwt_dwt2_fwd_pyramid.restype = ctypes.c_int
    lib.mifwt_dwt2_fwd_pyramid.argtypes = [ctypes.c_int, ctypes.POINTER(desc_p), vp, ctypes.POINTER(vpp), vp, dbl_p, dbl_p, vp]
    lib.mifwt_dwt2_inv_pair_supported.restype = ctypes.c_int
    lib.mifwt_dwt2_inv_pair_supported.argtypes = [desc_p, desc_p]
    lib.mifwt_dwt2_inv_pair.restype = ctypes.c_int
    lib.mifwt_dwt2_inv_pair.argtypes = [desc_p, desc_p, vp, ctypes.POINTER(vp), ctypes.POINTER(vp), vp, dbl_p, dbl_p, vp]
    lib.mifwt_dwt1_fwd_tail_max_n.restype = ctypes.c_int
    lib.mifwt_dwt1_fwd_tail_max_n.argtypes = [ctypes.c_int]
    lib.mifwt_dwt1_fwd_tail.restype = ctypes.c_int
    lib.mifwt_dwt1_fwd_tail.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, vp,
                                        ctypes.c_int64, vp, ctypes.c_int64, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_int64), dbl_p, dbl_p, vp]
    lib.mifwt_dwt1_fwd_long.restype = ctypes.c_int
    lib.mifwt_dwt1_fwd_long.argtypes = lib.mifwt_dwt1_fwd_tail.argtypes
    lib.mifwt_dwt1_fwd_long_levels.restype = ctypes.c_int
    lib.mifwt_dwt1_fwd_long_levels.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int]
    lib.mifwt_dwt1_inv_long_supported.restype = ctypes.c_int
    lib.mifwt_dwt1_inv_long_supported.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.POINTER(ctypes.c_int32)]
    lib.mifwt_dwt1_inv_long.restype = ctypes.c_int
    lib.mifwt_dwt1_inv_long.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.POINTER(ctypes.c_int32), vp, ctypes.c_int64,
                                        ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_int64), vp, ctypes.c_int64, dbl_p, dbl_p, vp]
    lib.mifwt_dwt1_inv_tail.restype = ctypes.c_int
    lib.mifwt_dwt1_inv_tail.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, vp, ctypes.c_int64,
                                        ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int32), vp,
                                        ctypes.c_int64, dbl_p, dbl_p, vp]
    experiment = mismatch  # (tools/: an older build accepted for a same-run comparison may lack the newest entry points)
    if not experiment or hasattr(lib, "mifwt_workspace_bytes_dtaps"):
        lib.mifwt_workspace_bytes_dtaps.restype = ctypes.c_size_t
        lib.mifwt_workspace_bytes_dtaps.argtypes = [desc_p, ctypes.c_int]
    if not experiment or hasattr(lib, "mifwt_kernel_id_dtaps"):  # (round 6; an older experiment build: level_events report id 0 for device taps)
        lib.mifwt_kernel_id_dtaps.restype = ctypes.c_int
        lib.mifwt_kernel_id_dtaps.argtypes = [desc_p, ctypes.c_int]
        for name in ("mifwt_dwt_fwd_dtaps", "mifwt_dwt_inv_dtaps", "mifwt_dwt_fwd_adjoint_dtaps", "mifwt_dwt_inv_adjoint_dtaps"):
            getattr(lib, name).restype = ctypes.c_int
        lib.mifwt_dwt_fwd_dtaps.argtypes = [desc_p, vp, vp, vpp, vp, vp, vp, ctypes.c_size_t, vp]
        lib.mifwt_dwt_inv_dtaps.argtypes = [desc_p, vp, vpp, vp, vp, vp, vp, ctypes.c_size_t, vp]
        lib.mifwt_dwt_fwd_adjoint_dtaps.argtypes = [desc_p, vp, vpp, vp, vp, vp, vp, ctypes.c_size_t, vp]
        lib.mifwt_dwt_inv_adjoint_dtaps.argtypes = [desc_p, vp, vp, vpp, vp, vp, vp, ctypes.c_size_t, vp]
    if not experiment or hasattr(lib, "mifwt_launch_count"):
        lib.mifwt_launch_count.restype = ctypes.c_uint64
        lib.mifwt_launch_count.argtypes = [ctypes.c_int]
    if not experiment or hasattr(lib, "mifwt_tap_correlate_planes"):  # (round 6)
        lib.mifwt_tap_correlate_planes.restype = ctypes.c_int
        lib.mifwt_tap_correlate_planes.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_int64] * 5 + [vp, ctypes.c_int64, ctypes.c_int64, vp, ctypes.c_int64,
                                                   ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]
        lib.mifwt_dwt1_inv_outer.restype = ctypes.c_int
        lib.mifwt_dwt1_inv_outer.argtypes = [ctypes.c_int] + [ctypes.c_int64] * 4 + [vp, ctypes.c_int64, ctypes.c_int64, vp, ctypes.c_int64, ctypes.c_int64, vp,
                                             ctypes.c_int64, ctypes.c_int64, ctypes.c_int, dbl_p, dbl_p, vp, vp, vp]
        lib.mifwt_dwt1_fwd_outer.restype = ctypes.c_int
        lib.mifwt_dwt1_fwd_outer.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, vp, ctypes.c_int64, ctypes.c_int64, vp, vp, ctypes.c_int64,
                                             ctypes.c_int64, ctypes.c_int, ctypes.c_int, dbl_p, dbl_p, vp, vp, vp]
    lib.mifwt_set_option.restype = ctypes.c_int
    lib.mifwt_set_option.argtypes = [ctypes.c_int, ctypes.c_int]
    _lib = lib
    return lib


def _check(rc: int) -> None:
    if rc != 0:
        msg = load_library().mifwt_strerror(rc).decode()
        raise RuntimeError(f"libmifwt: {msg} (code {rc})")


def _require_gpu(t: torch.Tensor) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"ptwt_amd: expected a tensor on a ROCm device, got device '{t.device}'. This engine runs on "
            "MI355X only; there is no CPU path."
        )


# Optional per-level device timing (bench.py's roofline leg): when set to a list, every level call
# appends (tag, kernel_id, start_event, end_event) recorded on the launch stream.
level_events: Optional[list] = None

ROW_ALIGN = int(os.environ.get("MIFWT_ROW_ALIGN", "1"))  # bytes; 1 = dense rows
# Row alignment (bytes) of the planes the streaming multi-level analysis kernel writes; 1 = dense.  MEASURED (round 4, config 2, same run,
# profiles/r04b_st16_ab.txt): rows padded to 16 bytes (515 -> 516 floats) cost 20 us per call with rotating output sets whatever the
# store width (126 us dense, 147-150 padded with 8-byte stores, 141-166 with 16-byte stores) — a memory-channel effect of the pitch,
# as on the 3-D planes.  Dense stays the default; the kernel's 16-byte store path serves planes whose dense rows are 16-byte aligned.
PYRAMID_ROW_ALIGN = int(os.environ.get("MIFWT_PYRAMID_ROW_ALIGN", "1"))

OPT_FORCE_GENERIC = 0
OPT_ROWS_PER_CHUNK = 1
OPT_PREFETCH_PAIRS = 2
OPT_NT_STORE = 4
OPT_TILE_MODE = 5
OPT_TILE_ROWS = 6
OPT_MFMA_MODE = 7
OPT_PAIR_MODE = 8
OPT_PAIR_ROWS = 9
OPT_DEBUG = 11
OPT_PYRAMID_MODE = 12
OPT_PYR_WGS = 13  # >0: kernel 16 cuts the batch's rows into this many chunks (tests of units that start / end anywhere)
OPT_EXP = 15  # experiment word of an A/B run (tools/)
KID_PAIR = 12
KID_INV_PAIR = 13
KID_TAIL = 14
KID_INV_TAIL = 15
KID_PYRAMID = 16
KID_LONG = 17
KID_INV_LONG = 18
KID_SMALL = 20
KID_INV_SMALL = 21
KID_INV_PYRAMID = 22
MAX_PYRAMID_LEVELS = 8  # mifwt_dwt2_fwd_pyramid: three for the streaming kernel, eight for the small-plane kernel


VARIANT_FWD_MFMA_WALK, VARIANT_FWD_MFMA_TILE, VARIANT_FWD_PYR_ST16, VARIANT_FWD_PYR_ST8 = 0, 1, 2, 3


def launch_count(variant: int) -> int:
    """Launches of a kernel variant enqueued by this process so far (variants share a kernel id; tests pin "this path ran" with it)."""
    return int(load_library().mifwt_launch_count(variant))


def set_option(key: int, value: int) -> None:
    """Library-wide test/diagnostic switches (e.g. ``OPT_FORCE_GENERIC`` to bypass the fused kernels)."""
    _check(load_library().mifwt_set_option(key, value))
    _plans.clear()  # cached plans hold the scratch size and kernel id of the routing that was in force
    for c in _routing_caches:
        c.clear()


_routing_caches: list = []  # other modules' per-geometry routing memos (cleared with the plans when an option changes)


class _Plan:
    """Everything about one level that depends only on geometry (extents, strides, dtype, mode, filter length):
    the filled ``mifwt_level_desc``, the output allocation, scratch size and kernel id.  Cached, so a repeated
    call costs one ``torch.empty`` + one C call per level on the host."""

    __slots__ = ("desc", "ref", "alloc_shape", "view_last", "nb", "plane_bytes", "ws_bytes", "kid", "empty")


_plans: dict = {}


def _trim_plans() -> None:
    """Keep the plan cache bounded without dropping everything at once: past 4096 entries the oldest quarter goes (dicts keep
    insertion order), so a caller that cycles through many geometries keeps its recent ones."""
    if len(_plans) > 4096:
        for k in list(_plans)[:1024]:
            _plans.pop(k, None)
_tls = threading.local()
_taps_cache: dict = {}


class DevTaps:
    """A filter of a bank that LIVES ON THE GPU (a learnable wavelet's parameter): float64, contiguous, L values.  Level methods that
    receive their taps as ``DevTaps`` call the ``mifwt_*_dtaps`` entry points — the kernels read the filter from device memory, nothing
    is copied to the host, nothing synchronises, the call can be captured into a HIP graph (include/mifwt.h)."""

    __slots__ = ("t",)

    def __init__(self, t: torch.Tensor):
        t = t.detach().reshape(-1)
        if t.dtype != torch.float64 or not t.is_contiguous():
            t = t.to(torch.float64).contiguous()  # (device-side cast, asynchronous)
        self.t = t

    def __len__(self) -> int:
        return int(self.t.shape[0])

    @property
    def ptr(self) -> int:
        return self.t.data_ptr()


def _is_dev(taps) -> bool:
    return isinstance(taps, DevTaps)


def _taps_array(taps: Sequence[float]):
    key = tuple(taps)
    arr = _taps_cache.get(key)
    if arr is None:
        if len(_taps_cache) > 512:
            _taps_cache.clear()
        arr = _taps_cache[key] = _arr(ctypes.c_double, len(key))(*key)
    return arr


def _band_ptrs(base: int, plane_bytes: int, n: int):
    """Device pointers of planes 1 .. n of a level buffer, in a fresh ctypes array: cached plans are shared between threads and
    ctypes releases the GIL during the C call, so a plan never owns an array that calls write to."""
    return _arr(ctypes.c_void_p, n)(*[base + s * plane_bytes for s in range(1, n + 1)])


def _raw_stream(dev_index: int) -> int:
    return torch._C._cuda_getCurrentRawStream(dev_index)


class HipLevelEngine:
    """One decomposition / reconstruction level for a folded batch, on the GPU, through the C ABI."""

    @staticmethod
    def _analysis_plan(x: torch.Tensor, flen: int, mode_id: int, min_align: int = 1) -> _Plan:
        lib = load_library()
        ndim = x.dim() - 1
        batch = x.shape[0]
        sig = [int(n) for n in x.shape[1:]]
        coef = [(n + 2 * ((2 * flen - 3) // 2) + (n % 2) - flen) // 2 + 1 for n in sig]
        nb = 1 << ndim
        # rows of the sub-band planes can be made to start on ROW_ALIGN-byte boundaries (pitch padded, the
        # returned bands are views of the padded buffer).  Measured on MI355X: no gain at 16 B, a loss at 128 B
        # (config 2), so the default is dense rows.
        # Exception: the matrix-core analysis kernel (f16 storage, 18..32 taps, 2-D) stores 16 bytes per lane; with the dense pitch of
        # an odd coefficient width every other row starts 2-byte aligned and the stores are split: level 1 of the config-5 slice
        # 4.3 ms dense, 3.7 ms with 16-byte, 2.76 ms with 128-byte aligned rows (tools/mfma_walk_parts.py) — those planes get 128.
        esz = x.element_size()
        align = max(ROW_ALIGN, min_align, 1)
        if align <= 1 and ndim == 2 and x.dtype == torch.float16 and 18 <= flen <= 32:
            align = 128
        pitch = -(-coef[-1] * esz // align) * align // esz if align > esz and ndim >= 2 else coef[-1]
        if min_align < 0 and ndim >= 2:  # (experiments, tools/pitch_sweep.py: -k = k extra elements per row, whatever the alignment)
            pitch = coef[-1] - min_align
        p = _Plan()
        p.alloc_shape = (batch, nb, *coef[:-1], pitch)
        p.view_last = coef[-1] if pitch != coef[-1] else None
        p.nb = nb
        p.empty = batch == 0 or min(coef) == 0
        d = LevelDesc()
        d.ndim, d.dtype, d.mode, d.filt_len, d.batch = ndim, _DTYPE_IDS[x.dtype], mode_id, flen, batch
        bstride = [1] * (ndim + 2)
        for i in range(ndim, -1, -1):
            bstride[i] = bstride[i + 1] * p.alloc_shape[i + 1]
        for a in range(ndim):
            d.sig_extent[a] = sig[a]
            d.coef_extent[a] = coef[a]
            d.sig_stride[1 + a] = x.stride(1 + a)
            d.approx_stride[1 + a] = d.detail_stride[1 + a] = bstride[2 + a]
        d.sig_stride[0] = x.stride(0)
        d.approx_stride[0] = d.detail_stride[0] = bstride[0]
        p.desc = d
        p.ref = ctypes.byref(d)
        p.plane_bytes = bstride[1] * esz
        p.ws_bytes = 0 if p.empty else lib.mifwt_workspace_bytes(p.ref, 0)
        p.kid = lib.mifwt_kernel_id(p.ref, 0)
        return p

    @staticmethod
    def _details_only(pl: _Plan) -> _Plan:
        """Copy of a level plan for a buffer WITHOUT the approximation plane ([B, 2^n - 1, M..]: plane s - 1 = band s): what a level of a
        multi-level launch gets whose approximation stays on chip (a [B, 2^n, M..] buffer would keep a dead plane alive as long as
        any of its detail bands lives).  The cached plan itself is shared with the per-level path and is not touched."""
        q = _Plan()
        d = LevelDesc()
        ctypes.memmove(ctypes.addressof(d), ctypes.addressof(pl.desc), ctypes.sizeof(LevelDesc))
        plane = pl.desc.detail_stride[0] // pl.nb
        d.detail_stride[0] = d.approx_stride[0] = plane * (pl.nb - 1)
        q.desc, q.ref = d, ctypes.byref(d)
        q.alloc_shape = (pl.alloc_shape[0], pl.nb - 1, *pl.alloc_shape[2:])
        q.view_last, q.nb, q.plane_bytes, q.ws_bytes, q.kid, q.empty = pl.view_last, pl.nb - 1, pl.plane_bytes, pl.ws_bytes, pl.kid, pl.empty
        return q

    def analysis(self, x: torch.Tensor, dec_lo: Sequence[float], dec_hi: Sequence[float], mode_id: int) -> torch.Tensor:
        """``x``: [B, N_0..N_{n-1}] (any strides) -> one buffer [B, 2^n, M_0..] whose plane ``s`` is band ``s``
        (bit (n-1-a) of s set <=> high-pass along axis a; plane 0 = approximation)."""
        _require_gpu(x)
        flen = len(dec_lo)
        key = (x.shape, x.stride(), x.dtype, mode_id, flen, ROW_ALIGN)
        p = _plans.get(key)
        if p is None:
            _trim_plans()
            p = _plans[key] = self._analysis_plan(x, flen, mode_id)
        buf = torch.empty(p.alloc_shape, dtype=x.dtype, device=x.device)
        if p.view_last is not None:
            buf = buf[..., : p.view_last]
        if p.empty:
            return buf
        base = buf.data_ptr()
        ptrs = _band_ptrs(base, p.plane_bytes, p.nb - 1)
        lib = _lib
        xp = x.data_ptr()
        if _is_dev(dec_lo):
            dl, dh = dec_lo.ptr, dec_hi.ptr
            self._run(p, 0, x, lambda ws, wsb, stream: lib.mifwt_dwt_fwd_dtaps(p.ref, xp, base, ptrs, dl, dh, ws, wsb, stream), kid=0, dtaps=True)
            return buf
        lo, hi = _taps_array(dec_lo), _taps_array(dec_hi)
        self._run(p, 0, x, lambda ws, wsb, stream: lib.mifwt_dwt_fwd(p.ref, xp, base, ptrs, lo, hi, ws, wsb, stream))
        return buf

    def analysis_pair(self, x: torch.Tensor, dec_lo: Sequence[float], dec_hi: Sequence[float], mode_id: int):
        """TWO consecutive 2-D analysis levels in one launch (C ABI ``mifwt_dwt2_fwd_pair``): ``x`` [B, H, W] ->
        ``(buf1, buf2)``: ``buf2`` laid out like an :meth:`analysis` result, ``buf1`` [B, 3, M, M] with the detail bands only (the intermediate
        approximation, which a pyramid does not return, stays on chip).  Returns None when the library does not
        serve this geometry as a pair; the caller then runs the levels one by one."""
        _require_gpu(x)
        if x.dim() != 3:
            return None
        flen = len(dec_lo)
        key = ("pair", x.shape, x.stride(), x.dtype, mode_id, flen, ROW_ALIGN)
        plan = _plans.get(key)
        if plan is None:
            _trim_plans()
            lib = load_library()
            p1 = self._analysis_plan(x, flen, mode_id)
            ok = False
            p2 = None
            if not p1.empty:
                # geometry of the level-2 call: its input is plane 0 of the level-1 buffer (strides only, no memory)
                lvl1 = torch.empty(p1.alloc_shape, dtype=x.dtype, device="meta")
                if p1.view_last is not None:
                    lvl1 = lvl1[..., : p1.view_last]
                p2 = self._analysis_plan(lvl1[:, 0], flen, mode_id)
                ok = (not p2.empty) and bool(lib.mifwt_dwt2_fwd_pair_supported(p1.ref, p2.ref))
                if ok:  # the first level's buffer: detail planes only (its approximation stays on chip)
                    lean = self._details_only(p1)
                    if lib.mifwt_dwt2_fwd_pair_supported(lean.ref, p2.ref):
                        p1 = lean
            plan = _plans[key] = (p1, p2, ok)
        p1, p2, ok = plan
        if not ok:
            return None
        buf1 = torch.empty(p1.alloc_shape, dtype=x.dtype, device=x.device)
        buf2 = torch.empty(p2.alloc_shape, dtype=x.dtype, device=x.device)
        if p1.view_last is not None:
            buf1 = buf1[..., : p1.view_last]
        if p2.view_last is not None:
            buf2 = buf2[..., : p2.view_last]
        b1, b2 = buf1.data_ptr(), buf2.data_ptr()
        ptrs1 = _band_ptrs(b1 - (4 - p1.nb) * p1.plane_bytes, p1.plane_bytes, 3)  # band ad: plane 1 of a full buffer, plane 0 of a details-only one
        ptrs2 = _band_ptrs(b2, p2.plane_bytes, 3)
        lo, hi = _taps_array(dec_lo), _taps_array(dec_hi)
        lib = _lib
        xp = x.data_ptr()
        self._run(p1, 0, x, lambda ws, wsb, stream: lib.mifwt_dwt2_fwd_pair(p1.ref, p2.ref, xp, ptrs1, b2, ptrs2, lo, hi, stream),
                  kid=KID_PAIR)
        return buf1, buf2

    def analysis_pyramid(self, x: torch.Tensor, dec_lo: Sequence[float], dec_hi: Sequence[float], mode_id: int, nlevels: int):
        """Several consecutive 2-D analysis levels in one launch (C ABI ``mifwt_dwt2_fwd_pyramid``: up to three through the streaming
        kernel, up to eight — the whole pyramid — for planes that fit into LDS): ``x`` [B, H, W] -> a list of
        buffers, finest first: the LAST one laid out like an :meth:`analysis` result ([B, 4, M, M]: approximation + three detail bands),
        the others [B, 3, M, M] with the detail bands only (ad, da, dd) — their approximations never leave the chip.  Fuses as many of the ``nlevels`` requested levels as the library
        serves for this geometry (possibly fewer); returns None when it serves none."""
        _require_gpu(x)
        if x.dim() != 3 or x.dtype != torch.float32:
            return None
        flen = len(dec_lo)
        key, plan = self._pyramid_plan(x, flen, mode_id, nlevels)
        plans, n_ok, refs, kid = plan
        if n_ok == 0:
            return None
        bufs = []
        for pl in plans:
            b = torch.empty(pl.alloc_shape, dtype=x.dtype, device=x.device)
            bufs.append(b if pl.view_last is None else b[..., : pl.view_last])
        # the band-pointer arrays are per thread: cached plans are shared between threads, and ctypes drops the GIL in the call
        skey = (key, n_ok)  # (a routing option may change how many levels the same geometry fuses)
        slot = _tls.__dict__.setdefault("pyr", {}).get(skey)
        if slot is None:
            rows = [_arr(ctypes.c_void_p, 3)() for _ in plans]
            det = _arr(ctypes.POINTER(ctypes.c_void_p), n_ok)(*[ctypes.cast(r, ctypes.POINTER(ctypes.c_void_p)) for r in rows])
            slot = _tls.pyr[skey] = (rows, det)
            if len(_tls.pyr) > 256:
                _tls.pyr.clear()
                _tls.pyr[skey] = slot
        rows, det = slot
        for r, b, pl in zip(rows, bufs, plans):
            pb = pl.plane_bytes
            base = b.data_ptr() + (pl.nb - 3) * pb  # band ad: plane 1 of a full buffer, plane 0 of a details-only one
            r[0], r[1], r[2] = base, base + pb, base + 2 * pb
        lo, hi = _taps_array(dec_lo), _taps_array(dec_hi)
        lib = _lib
        xp, ap = x.data_ptr(), bufs[-1].data_ptr()
        self._run(plans[0], 0, x, lambda ws, wsb, stream: lib.mifwt_dwt2_fwd_pyramid(n_ok, refs, xp, det, ap, lo, hi, stream), kid=kid)
        return bufs

    def pyramid_levels(self, x: torch.Tensor, flen: int, mode_id: int, nlevels: int) -> int:
        """How many of the next ``nlevels`` 2-D analysis levels :meth:`analysis_pyramid` would take in one launch for this geometry
        (0: none); nothing is launched."""
        if x.dim() != 3 or x.dtype != torch.float32 or not x.is_cuda:
            return 0
        return self._pyramid_plan(x, flen, mode_id, nlevels)[1][1]

    def _pyramid_plan(self, x: torch.Tensor, flen: int, mode_id: int, nlevels: int):
        """(cache key, (level plans, levels served, descriptor array, kernel id)) of :meth:`analysis_pyramid` for a geometry."""
        key = ("pyr", x.shape, x.stride(), mode_id, flen, min(nlevels, MAX_PYRAMID_LEVELS), ROW_ALIGN, PYRAMID_ROW_ALIGN)
        plan = _plans.get(key)
        if plan is None:
            _trim_plans()
            lib = load_library()

            def chain(min_align):
                plans = [self._analysis_plan(x, flen, mode_id, min_align)]
                while len(plans) < min(nlevels, MAX_PYRAMID_LEVELS) and not plans[-1].empty:
                    pl = plans[-1]
                    lvl = torch.empty(pl.alloc_shape, dtype=x.dtype, device="meta")
                    if pl.view_last is not None:
                        lvl = lvl[..., : pl.view_last]
                    plans.append(self._analysis_plan(lvl[:, 0], flen, mode_id, min_align))
                n_ok, route = 0, 0
                if not any(pl.empty for pl in plans):
                    for n in range(len(plans), 0, -1):
                        refs = (ctypes.POINTER(LevelDesc) * n)(*[ctypes.pointer(pl.desc) for pl in plans[:n]])
                        route = lib.mifwt_dwt2_fwd_pyramid_supported(n, refs)
                        if route:
                            n_ok = n
                            break
                return plans, n_ok, route

            plans, n_ok, route = chain(1)
            if n_ok and route == 1 and (PYRAMID_ROW_ALIGN > 1 or PYRAMID_ROW_ALIGN < 0):
                # the streaming kernel stores 16 bytes per lane when the rows of every plane it writes start on 16-byte boundaries
                # (lane pairs exchange rows in front of the store, csrc/mifwt_pyr.h): its planes get a row pitch of a multiple of four
                # floats (config 2: 515 -> 516); the returned bands are views with that pitch.  The small-plane kernel (route 2) keeps
                # dense planes.
                plans_a, n_a, route_a = chain(PYRAMID_ROW_ALIGN)
                if n_a == n_ok and route_a == route:
                    plans = plans_a
            keep = plans[:n_ok]
            refs = (ctypes.POINTER(LevelDesc) * n_ok)(*[ctypes.pointer(pl.desc) for pl in keep]) if n_ok else None
            if n_ok > 1:  # every level but the last: detail planes only
                lean = [self._details_only(pl) for pl in keep[:-1]] + [keep[-1]]
                lrefs = (ctypes.POINTER(LevelDesc) * n_ok)(*[ctypes.pointer(pl.desc) for pl in lean])
                if lib.mifwt_dwt2_fwd_pyramid_supported(n_ok, lrefs) == route:
                    keep, refs = lean, lrefs
            if n_ok:
                keep[0].ws_bytes = 0  # (the multi-level launches need no scratch; the plan's figure is that of the per-level route)
            plan = _plans[key] = (keep, n_ok, refs, KID_SMALL if route == 2 else KID_PYRAMID)
        return key, plan

    def analysis_tail(self, x: torch.Tensor, dec_lo: Sequence[float], dec_hi: Sequence[float], mode_id: int, nlevels: int):
        """The next levels of a 1-D decomposition in ONE launch — all ``nlevels`` remaining ones once a row fits into a workgroup (C
        ABI ``mifwt_dwt1_fwd_tail``), as many as the chunked long-row kernel fuses before that (``mifwt_dwt1_fwd_long``): ``x`` [B, N]
        -> a list of up to ``nlevels`` buffers, finest first: [B, 1, M_l] holding that level's detail coefficients, and for the LAST level
        [B, 2, M] laid out like an :meth:`analysis` result (plane 0 = its approximation, plane 1 = its details); the approximations in
        between never leave the chip.  Returns None outside the kernel's envelope."""
        _require_gpu(x)
        if x.dim() != 2 or x.dtype not in (torch.float32, torch.float64) or x.stride(1) != 1 or nlevels < 2 or nlevels > 24:
            return None
        lib = load_library()
        flen = len(dec_lo)
        rows, n0 = x.shape
        if rows == 0 or n0 == 0 or flen > 32:
            return None
        # rows too long for one workgroup, or too few rows to occupy the chip with one workgroup each: the chunked kernel fuses
        # as many levels as its halo rule allows (C ABI mifwt_dwt1_fwd_long), the caller comes back for the rest
        k_long = lib.mifwt_dwt1_fwd_long_levels(_DTYPE_IDS[x.dtype], flen, mode_id, rows, n0, nlevels) if x.dtype == torch.float32 else 0
        long_rows = k_long >= 2
        if long_rows:
            nlevels = k_long
        elif n0 > lib.mifwt_dwt1_fwd_tail_max_n(_DTYPE_IDS[x.dtype]):
            return None
        sizes, n = [], n0
        for _ in range(nlevels):
            n = (n + flen - 1) // 2
            sizes.append(n)
        # the last level's buffer carries the approximation in plane 0; the others hold their detail row only (their approximations
        # never leave the chip: a [B, 2, M] buffer would keep as many dead bytes alive as the coefficients themselves)
        last = nlevels - 1
        bufs = [torch.empty((rows, 2 if i == last else 1, m), dtype=x.dtype, device=x.device) for i, m in enumerate(sizes)]
        esz = x.element_size()
        det = _arr(ctypes.c_void_p, nlevels)(*[b.data_ptr() + (sizes[i] * esz if i == last else 0) for i, b in enumerate(bufs)])
        det_rs = _arr(ctypes.c_int64, nlevels)(*[(2 if i == last else 1) * m for i, m in enumerate(sizes)])
        lo, hi = _taps_array(dec_lo), _taps_array(dec_hi)
        p = _Plan()
        p.ws_bytes, p.kid = 0, (KID_LONG if long_rows else KID_TAIL)
        d = LevelDesc()
        d.ndim = 1
        d.sig_extent[0] = n0
        p.desc = d
        xp, ap = x.data_ptr(), bufs[-1].data_ptr()
        rc_box = []

        def call(ws, wsb, stream):
            entry = lib.mifwt_dwt1_fwd_long if long_rows else lib.mifwt_dwt1_fwd_tail
            rc = entry(_DTYPE_IDS[x.dtype], flen, mode_id, rows, n0, nlevels, xp, x.stride(0), ap, 2 * sizes[-1], det, det_rs, lo, hi, stream)
            rc_box.append(rc)
            return 0 if rc == -2 else rc  # "unsupported" is an answer here, not an error

        self._run(p, 0, x, call)
        if rc_box and rc_box[0] == -2:
            return None
        return bufs

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
        if any(t.stride() != ref_stride for t in details):
            details = [t.contiguous() for t in details]
            ref_stride = details[0].stride()
        key = ("inv", approx.shape, approx.stride(), ref_stride, approx.dtype, flen, tuple(out_extent))
        p = _plans.get(key)
        if p is None:
            _trim_plans()
            p = _Plan()
            d = LevelDesc()
            d.ndim, d.dtype, d.mode, d.filt_len, d.batch = ndim, _DTYPE_IDS[approx.dtype], 0, flen, batch
            for a in range(ndim):
                d.sig_extent[a] = int(out_extent[a])
                d.coef_extent[a] = int(approx.shape[1 + a])
            for a in range(ndim + 1):
                d.sig_stride[a] = y.stride(a)
                d.approx_stride[a] = approx.stride(a)
                d.detail_stride[a] = ref_stride[a]
            p.desc = d
            p.ref = ctypes.byref(d)
            p.ws_bytes = lib.mifwt_workspace_bytes(p.ref, 1)
            p.kid = lib.mifwt_kernel_id(p.ref, 1)
            _plans[key] = p
        ptrs = _arr(ctypes.c_void_p, len(details))(*[t.data_ptr() for t in details])
        ap, yp = approx.data_ptr(), y.data_ptr()
        if _is_dev(rec_lo):
            dl, dh = rec_lo.ptr, rec_hi.ptr
            self._run(p, 1, approx, lambda ws, wsb, stream: lib.mifwt_dwt_inv_dtaps(p.ref, ap, ptrs, yp, dl, dh, ws, wsb, stream), kid=0, dtaps=True)
            return y
        lo, hi = _taps_array(rec_lo), _taps_array(rec_hi)
        self._run(p, 1, approx, lambda ws, wsb, stream: lib.mifwt_dwt_inv(p.ref, ap, ptrs, yp, lo, hi, ws, wsb, stream))
        return y

    def synthesis_tail(self, approx: torch.Tensor, details: List[torch.Tensor], rec_lo: Sequence[float], rec_hi: Sequence[float],
                       out_lens: Sequence[int]):
        """The first ``len(details)`` (coarsest) levels of a 1-D reconstruction in ONE launch (C ABI ``mifwt_dwt1_inv_tail``):
        ``approx`` [B, m], ``details[l]`` [B, m_l] coarsest first, ``out_lens[l]`` the (already trimmed) output length of level l
        -> y [B, out_lens[-1]].  Returns None outside the kernel's envelope."""
        _require_gpu(approx)
        nl = len(details)
        if approx.dim() != 2 or approx.dtype not in (torch.float32, torch.float64) or nl < 2 or nl > 24:
            return None
        lib = load_library()
        flen = len(rec_lo)
        rows, m0 = approx.shape
        cap = lib.mifwt_dwt1_fwd_tail_max_n(_DTYPE_IDS[approx.dtype])
        if rows == 0 or m0 == 0 or flen > 32 or m0 > cap or max(out_lens) > cap or min(out_lens) < 1:
            return None
        if approx.stride(1) != 1:
            approx = approx.contiguous()
        details = [t if t.stride(1) == 1 else t.contiguous() for t in details]
        y = torch.empty((rows, int(out_lens[-1])), dtype=approx.dtype, device=approx.device)
        det = _arr(ctypes.c_void_p, nl)(*[t.data_ptr() for t in details])
        det_rs = _arr(ctypes.c_int64, nl)(*[t.stride(0) for t in details])
        outs = (ctypes.c_int32 * nl)(*[int(v) for v in out_lens])
        lo, hi = _taps_array(rec_lo), _taps_array(rec_hi)
        p = _Plan()
        p.ws_bytes, p.kid = 0, KID_INV_TAIL
        d = LevelDesc()
        d.ndim = 1
        d.sig_extent[0] = int(out_lens[-1])
        p.desc = d
        ap, yp = approx.data_ptr(), y.data_ptr()
        rc_box = []

        def call(ws, wsb, stream):
            rc = lib.mifwt_dwt1_inv_tail(_DTYPE_IDS[approx.dtype], flen, rows, m0, nl, ap, approx.stride(0), det, det_rs, outs, yp,
                                         y.stride(0), lo, hi, stream)
            rc_box.append(rc)
            return 0 if rc == -2 else rc  # "unsupported" is an answer here, not an error

        self._run(p, 1, approx, call)
        if rc_box and rc_box[0] == -2:
            return None
        return y

    def synthesis_long(self, approx: torch.Tensor, details: List[torch.Tensor], rec_lo: Sequence[float], rec_hi: Sequence[float],
                       out_lens: Sequence[int]):
        """The FINEST levels of a 1-D reconstruction in one launch, a chunk of the output row per workgroup (C ABI
        ``mifwt_dwt1_inv_long``): same arguments as :meth:`synthesis_tail`.  Fuses as many of the given levels as the kernel's halo
        rule allows, counted from the finest: returns ``(y, n_fused)`` with ``y`` the output of the last given level when all of
        them were fused — or ``(None, k)`` telling the caller to run the first ``len(details) - k`` levels some other way first and
        come back; ``(None, 0)`` outside the kernel's envelope."""
        _require_gpu(approx)
        nl = len(details)
        if approx.dim() != 2 or approx.dtype not in (torch.float32, torch.float64) or nl < 2:
            return None, 0
        lib = load_library()
        flen = len(rec_lo)
        rows = approx.shape[0]
        lens = [int(approx.shape[1])] + [int(v) for v in out_lens]
        k = min(nl, 8)
        while k >= 2:
            m = (ctypes.c_int32 * (k + 1))(*lens[nl - k:])
            if lib.mifwt_dwt1_inv_long_supported(_DTYPE_IDS[approx.dtype], flen, rows, k, m):
                break
            k -= 1
        if k < 2:
            return None, 0
        if k < nl:
            return None, k
        if approx.stride(1) != 1:
            approx = approx.contiguous()
        details = [t if t.stride(1) == 1 else t.contiguous() for t in details]
        y = torch.empty((rows, lens[-1]), dtype=approx.dtype, device=approx.device)
        det = _arr(ctypes.c_void_p, nl)(*[t.data_ptr() for t in details])
        det_rs = _arr(ctypes.c_int64, nl)(*[t.stride(0) for t in details])
        lo, hi = _taps_array(rec_lo), _taps_array(rec_hi)
        p = _Plan()
        p.ws_bytes, p.kid = 0, KID_INV_LONG
        d = LevelDesc()
        d.ndim = 1
        d.sig_extent[0] = lens[-1]
        p.desc = d
        ap, yp = approx.data_ptr(), y.data_ptr()
        self._run(p, 1, approx, lambda ws, wsb, stream: lib.mifwt_dwt1_inv_long(_DTYPE_IDS[approx.dtype], flen, rows, nl, m, ap, approx.stride(0), det, det_rs, yp,
                                                                              y.stride(0), lo, hi, stream))
        return y, nl

    def synthesis_pair(self, approx2: torch.Tensor, details2: List[torch.Tensor], details1: List[torch.Tensor],
                       rec_lo: Sequence[float], rec_hi: Sequence[float], out_extent: Sequence[int]):
        """TWO consecutive 2-D synthesis levels in one launch (C ABI ``mifwt_dwt2_inv_pair``): the coarser level's bands
        ``approx2`` / ``details2`` [B, M2h, M2w], the finer level's ``details1`` [B, M1h, M1w] (whose extents are the cropped
        output extents of the coarser level) -> y [B, *out_extent].  Returns None when the library does not serve this
        geometry as a pair; the caller then runs the levels one by one."""
        _require_gpu(approx2)
        if approx2.dim() != 3 or approx2.dtype != torch.float32:
            return None
        lib = load_library()
        flen = len(rec_lo)
        batch = approx2.shape[0]
        st2 = details2[0].stride()
        if any(t.stride() != st2 for t in details2):
            details2 = [t.contiguous() for t in details2]
            st2 = details2[0].stride()
        st1 = details1[0].stride()
        if any(t.stride() != st1 for t in details1):
            details1 = [t.contiguous() for t in details1]
            st1 = details1[0].stride()
        m1 = tuple(details1[0].shape[1:])
        key = ("invpair", approx2.shape, approx2.stride(), st2, m1, st1, flen, tuple(out_extent))
        plan = _plans.get(key)
        if plan is None:
            _trim_plans()
            d2, d1 = LevelDesc(), LevelDesc()
            for d in (d1, d2):
                d.ndim, d.dtype, d.mode, d.filt_len, d.batch = 2, _DTYPE_IDS[approx2.dtype], 0, flen, batch
            for a in range(2):
                d2.sig_extent[a] = int(m1[a])
                d2.coef_extent[a] = int(approx2.shape[1 + a])
                d1.sig_extent[a] = int(out_extent[a])
                d1.coef_extent[a] = int(m1[a])
            ydense = [int(out_extent[0]) * int(out_extent[1]), int(out_extent[1]), 1]
            lldense = [int(m1[0]) * int(m1[1]), int(m1[1]), 1]
            for a in range(3):
                d2.sig_stride[a] = lldense[a]       # never materialised
                d2.approx_stride[a] = approx2.stride(a)
                d2.detail_stride[a] = st2[a]
                d1.sig_stride[a] = ydense[a]
                d1.approx_stride[a] = lldense[a]    # ignored
                d1.detail_stride[a] = st1[a]
            p = _Plan()
            p.desc = d1
            p.ref = ctypes.byref(d1)
            p.ws_bytes = 0
            p.kid = KID_INV_PAIR
            ok = bool(lib.mifwt_dwt2_inv_pair_supported(ctypes.byref(d2), p.ref))
            plan = _plans[key] = (p, d2, ctypes.byref(d2), ok)
        p, _d2, ref2, ok = plan
        if not ok:
            return None
        y = torch.empty((batch, *out_extent), dtype=approx2.dtype, device=approx2.device)
        ptrs2 = _arr(ctypes.c_void_p, 3)(*[t.data_ptr() for t in details2])  # per call: plans are shared between threads
        ptrs1 = _arr(ctypes.c_void_p, 3)(*[t.data_ptr() for t in details1])
        lo, hi = _taps_array(rec_lo), _taps_array(rec_hi)
        ap, yp = approx2.data_ptr(), y.data_ptr()
        self._run(p, 1, approx2, lambda ws, wsb, stream: lib.mifwt_dwt2_inv_pair(ref2, p.ref, ap, ptrs2, ptrs1, yp, lo, hi, stream))
        return y

    def synthesis_pyramid_plan(self, approx: torch.Tensor, levels: List[List[torch.Tensor]], flen: int, out_extent: Sequence[int]):
        """The cached plan of :meth:`synthesis_pyramid` for this geometry — ``(plan, descs, refs, route)`` with route 0 (the library does
        not serve it as one launch), 1 (every level of a small plane, kernel id 21) or 2 (the up-to-three levels handed over of a big
        plane, kernel id 22).  Geometry only: ``approx`` and the bands may be meta tensors."""
        n = len(levels)
        if approx.dim() != 3 or approx.dtype != torch.float32 or n < 1 or n > MAX_PYRAMID_LEVELS:
            return None
        batch = approx.shape[0]
        key = ("invpyr", approx.shape, approx.stride(), tuple((lv[0].shape, lv[0].stride()) for lv in levels), flen, tuple(out_extent))
        plan = _plans.get(key)
        if plan is None:
            _trim_plans()
            lib = load_library()
            descs = []
            for i, lv in enumerate(levels):
                d = LevelDesc()
                d.ndim, d.dtype, d.mode, d.filt_len, d.batch = 2, _DTYPE_IDS[approx.dtype], 0, flen, batch
                m = lv[0].shape[1:]
                out = levels[i + 1][0].shape[1:] if i + 1 < n else out_extent
                for a in range(2):
                    d.coef_extent[a] = int(m[a])
                    d.sig_extent[a] = int(out[a])
                dense = [int(m[0]) * int(m[1]), int(m[1]), 1]
                ydense = [int(out[0]) * int(out[1]), int(out[1]), 1]
                for a in range(3):
                    d.sig_stride[a] = ydense[a]
                    d.approx_stride[a] = approx.stride(a) if i == 0 else dense[a]
                    d.detail_stride[a] = lv[0].stride(a)
                descs.append(d)
            refs = (ctypes.POINTER(LevelDesc) * n)(*[ctypes.pointer(d) for d in descs])
            route = int(lib.mifwt_dwt2_inv_pyramid_supported(n, refs)) if tuple(approx.shape[1:]) == tuple(levels[0][0].shape[1:]) else 0
            p = _Plan()
            p.desc = descs[-1]
            p.ref = ctypes.byref(descs[-1])
            p.ws_bytes = 0
            p.kid = KID_INV_SMALL if route == 1 else KID_INV_PYRAMID
            plan = _plans[key] = (p, descs, refs, route)
        return plan

    def synthesis_pyramid(self, approx: torch.Tensor, levels: List[List[torch.Tensor]], rec_lo: Sequence[float],
                          rec_hi: Sequence[float], out_extent: Sequence[int], plan=None):
        """Several levels of a 2-D reconstruction in one launch (C ABI ``mifwt_dwt2_inv_pyramid``): EVERY level of a small plane (kernel
        id 21), or the up-to-three levels handed over of a big one (kernel id 22, rows streamed through LDS rings).  ``approx``: the
        coarsest approximation [B, Mh, Mw], ``levels`` = per level (coarsest first) its bands ad, da, dd [B, Mh_l, Mw_l]; the running
        approximation is cropped to the next level's band extents, the finest level's output to ``out_extent``.  ``plan``: what
        :meth:`synthesis_pyramid_plan` returned for this very geometry (saves the lookup).
        Returns y [B, *out_extent], or None when the library does not serve this geometry (the caller then goes level by level)."""
        _require_gpu(approx)
        if plan is None:
            plan = self.synthesis_pyramid_plan(approx, levels, len(rec_lo), out_extent)
            if plan is None:
                return None
        p, _descs, refs, route = plan
        if not route:
            return None
        # the three detail bands of a level share their strides (views into one level buffer do; three separate dense tensors do);
        # anything else goes level by level
        for lv in levels:
            want = lv[0].stride()
            if lv[1].stride() != want or lv[2].stride() != want:
                return None
        n = len(levels)
        y = torch.empty((approx.shape[0], *out_extent), dtype=approx.dtype, device=approx.device)
        # the band-pointer arrays are per thread and reused (cached plans are shared between threads, and ctypes drops the GIL in the
        # call; fresh ctypes arrays + casts on every call are reference cycles that the garbage collector has to find: a 35 ms
        # pause every few hundred calls, tools/host_bound.py)
        slots = _tls.__dict__.setdefault("invpyr", {})
        slot = slots.get(n)
        if slot is None:
            rows = [_arr(ctypes.c_void_p, 3)() for _ in range(n)]
            det = _arr(ctypes.POINTER(ctypes.c_void_p), n)(*[ctypes.cast(r, ctypes.POINTER(ctypes.c_void_p)) for r in rows])
            slot = slots[n] = (rows, det)
        rows, det = slot
        for r, lv in zip(rows, levels):
            r[0], r[1], r[2] = lv[0].data_ptr(), lv[1].data_ptr(), lv[2].data_ptr()
        lo, hi = _taps_array(rec_lo), _taps_array(rec_hi)
        lib = _lib
        ap, yp = approx.data_ptr(), y.data_ptr()
        self._run(p, 1, approx, lambda ws, wsb, stream: lib.mifwt_dwt2_inv_pyramid(n, refs, ap, det, yp, lo, hi, stream))
        return y

    # ---- adjoints (reverse-mode differentiation; C ABI mifwt_dwt_fwd_adjoint / mifwt_dwt_inv_adjoint) -------------
    def analysis_adjoint(self, g_buf: torch.Tensor, sig_shape: Sequence[int], dec_lo: Sequence[float],
                         dec_hi: Sequence[float], mode_id: int) -> torch.Tensor:
        """Transpose of :meth:`analysis`: ``g_buf`` [B, 2^n, M_0..] (gradient of the level buffer) -> gradient of
        the level input, dense [B, *sig_shape]."""
        _require_gpu(g_buf)
        lib = load_library()
        if g_buf.stride(-1) != 1:
            g_buf = g_buf.contiguous()
        ndim = g_buf.dim() - 2
        flen = len(dec_lo)
        batch = g_buf.shape[0]
        g_x = torch.empty((batch, *sig_shape), dtype=g_buf.dtype, device=g_buf.device)
        if g_x.numel() == 0:
            return g_x
        key = ("fwd_adj", g_buf.shape, g_buf.stride(), tuple(sig_shape), g_buf.dtype, mode_id, flen)
        p = _plans.get(key)
        if p is None:
            p = _Plan()
            d = LevelDesc()
            d.ndim, d.dtype, d.mode, d.filt_len, d.batch = ndim, _DTYPE_IDS[g_buf.dtype], mode_id, flen, batch
            for a in range(ndim):
                d.sig_extent[a] = int(sig_shape[a])
                d.coef_extent[a] = int(g_buf.shape[2 + a])
                d.sig_stride[1 + a] = g_x.stride(1 + a)
                d.approx_stride[1 + a] = d.detail_stride[1 + a] = g_buf.stride(2 + a)
            d.sig_stride[0] = g_x.stride(0)
            d.approx_stride[0] = d.detail_stride[0] = g_buf.stride(0)
            p.desc, p.ref = d, ctypes.byref(d)
            p.nb = 1 << ndim
            p.plane_bytes = g_buf.stride(1) * g_buf.element_size()
            p.ws_bytes = lib.mifwt_workspace_bytes(p.ref, 2)
            p.kid = lib.mifwt_kernel_id(p.ref, 2)
            _plans[key] = p
        base = g_buf.data_ptr()
        ptrs = _band_ptrs(base, p.plane_bytes, p.nb - 1)
        gp = g_x.data_ptr()
        if _is_dev(dec_lo):
            dl, dh = dec_lo.ptr, dec_hi.ptr
            self._run(p, 2, g_buf, lambda ws, wsb, stream: lib.mifwt_dwt_fwd_adjoint_dtaps(p.ref, base, ptrs, gp, dl, dh, ws, wsb, stream), kid=0, dtaps=True)
            return g_x
        lo, hi = _taps_array(dec_lo), _taps_array(dec_hi)
        self._run(p, 2, g_buf, lambda ws, wsb, stream: lib.mifwt_dwt_fwd_adjoint(p.ref, base, ptrs, gp, lo, hi, ws, wsb, stream))
        return g_x

    def analysis_adjoint_bands(self, g_approx: torch.Tensor, g_details: Sequence[torch.Tensor], sig_shape: Sequence[int], dec_lo: Sequence[float],
                               dec_hi: Sequence[float], mode_id: int) -> torch.Tensor:
        """:meth:`analysis_adjoint` with the gradient of every band in a tensor of its own, ``g_approx`` and the 2^n - 1 ``g_details``
        [B, M_0..] each (what the backward of a multi-level launch is handed: the approximation's gradient is the result of the coarser
        level's adjoint, the details' gradients come from the caller one by one) — no concatenation; the C ABI takes a pointer per band."""
        _require_gpu(g_approx)
        lib = load_library()
        if g_approx.stride(-1) != 1:
            g_approx = g_approx.contiguous()
        ref_stride = g_details[0].stride()
        if ref_stride[-1] != 1 or any(t.stride() != ref_stride for t in g_details):
            g_details = [t.contiguous() for t in g_details]
        ndim = g_approx.dim() - 1
        flen = len(dec_lo)
        batch = g_approx.shape[0]
        g_x = torch.empty((batch, *sig_shape), dtype=g_approx.dtype, device=g_approx.device)
        if g_x.numel() == 0:
            return g_x
        gd0 = g_details[0]
        key = ("fwd_adjb", g_approx.shape, g_approx.stride(), gd0.stride(), tuple(sig_shape), g_approx.dtype, mode_id, flen)
        p = _plans.get(key)
        if p is None:
            p = _Plan()
            d = LevelDesc()
            d.ndim, d.dtype, d.mode, d.filt_len, d.batch = ndim, _DTYPE_IDS[g_approx.dtype], mode_id, flen, batch
            for a in range(ndim):
                d.sig_extent[a] = int(sig_shape[a])
                d.coef_extent[a] = int(g_approx.shape[1 + a])
                d.sig_stride[1 + a] = g_x.stride(1 + a)
                d.approx_stride[1 + a] = g_approx.stride(1 + a)
                d.detail_stride[1 + a] = gd0.stride(1 + a)
            d.sig_stride[0] = g_x.stride(0)
            d.approx_stride[0] = g_approx.stride(0)
            d.detail_stride[0] = gd0.stride(0)
            p.desc, p.ref = d, ctypes.byref(d)
            p.nb = 1 << ndim
            p.ws_bytes = lib.mifwt_workspace_bytes(p.ref, 2)
            p.kid = lib.mifwt_kernel_id(p.ref, 2)
            _plans[key] = p
        ptrs = _arr(ctypes.c_void_p, p.nb - 1)(*[t.data_ptr() for t in g_details])
        ap, gp = g_approx.data_ptr(), g_x.data_ptr()
        if _is_dev(dec_lo):
            dl, dh = dec_lo.ptr, dec_hi.ptr
            self._run(p, 2, g_approx, lambda ws, wsb, stream: lib.mifwt_dwt_fwd_adjoint_dtaps(p.ref, ap, ptrs, gp, dl, dh, ws, wsb, stream), kid=0, dtaps=True)
            return g_x
        lo, hi = _taps_array(dec_lo), _taps_array(dec_hi)
        self._run(p, 2, g_approx, lambda ws, wsb, stream: lib.mifwt_dwt_fwd_adjoint(p.ref, ap, ptrs, gp, lo, hi, ws, wsb, stream))
        return g_x

    def synthesis_adjoint(self, g_y: torch.Tensor, coef_shape: Sequence[int], rec_lo: Sequence[float],
                          rec_hi: Sequence[float]) -> torch.Tensor:
        """Transpose of :meth:`synthesis`: ``g_y`` [B, *out_extent] -> one buffer [B, 2^n, *coef_shape] whose plane
        ``s`` is the gradient of band ``s`` (plane 0: the approximation)."""
        _require_gpu(g_y)
        lib = load_library()
        if g_y.stride(-1) != 1:
            g_y = g_y.contiguous()
        ndim = g_y.dim() - 1
        flen = len(rec_lo)
        batch = g_y.shape[0]
        nb = 1 << ndim
        g_buf = torch.empty((batch, nb, *coef_shape), dtype=g_y.dtype, device=g_y.device)
        if g_buf.numel() == 0:
            return g_buf
        key = ("inv_adj", g_y.shape, g_y.stride(), tuple(coef_shape), g_y.dtype, flen)
        p = _plans.get(key)
        if p is None:
            p = _Plan()
            d = LevelDesc()
            d.ndim, d.dtype, d.mode, d.filt_len, d.batch = ndim, _DTYPE_IDS[g_y.dtype], 0, flen, batch
            for a in range(ndim):
                d.sig_extent[a] = int(g_y.shape[1 + a])
                d.coef_extent[a] = int(coef_shape[a])
                d.sig_stride[1 + a] = g_y.stride(1 + a)
                d.approx_stride[1 + a] = d.detail_stride[1 + a] = g_buf.stride(2 + a)
            d.sig_stride[0] = g_y.stride(0)
            d.approx_stride[0] = d.detail_stride[0] = g_buf.stride(0)
            p.desc, p.ref = d, ctypes.byref(d)
            p.nb = nb
            p.plane_bytes = g_buf.stride(1) * g_buf.element_size()
            p.ws_bytes = lib.mifwt_workspace_bytes(p.ref, 3)
            p.kid = lib.mifwt_kernel_id(p.ref, 3)
            _plans[key] = p
        base = g_buf.data_ptr()
        ptrs = _band_ptrs(base, p.plane_bytes, nb - 1)
        yp = g_y.data_ptr()
        if _is_dev(rec_lo):
            dl, dh = rec_lo.ptr, rec_hi.ptr
            self._run(p, 3, g_y, lambda ws, wsb, stream: lib.mifwt_dwt_inv_adjoint_dtaps(p.ref, yp, base, ptrs, dl, dh, ws, wsb, stream), kid=0, dtaps=True)
            return g_buf
        lo, hi = _taps_array(rec_lo), _taps_array(rec_hi)
        self._run(p, 3, g_y, lambda ws, wsb, stream: lib.mifwt_dwt_inv_adjoint(p.ref, yp, base, ptrs, lo, hi, ws, wsb, stream))
        return g_buf

    def tap_correlate(self, a: torch.Tensor, b: torch.Tensor, filt_len: int, c0: int, sgn: int, mode_id: int,
                      out: torch.Tensor) -> None:
        """``out[t] += sum_{row, k} a[row, k] * b_ext[row, 2k + c0 + sgn t]`` (C ABI ``mifwt_tap_correlate``): ``a`` [rows, M],
        ``b`` [rows, N] (contiguous samples), ``out`` float64 [filt_len] on the same device, accumulated into."""
        _require_gpu(a)
        lib = load_library()
        if a.stride(-1) != 1:
            a = a.contiguous()
        if b.stride(-1) != 1:
            b = b.contiguous()
        assert a.dim() == 2 and b.dim() == 2 and a.shape[0] == b.shape[0] and out.dtype == torch.float64
        with torch.cuda.device(a.device):
            rc = lib.mifwt_tap_correlate(_DTYPE_IDS[a.dtype], a.shape[0], a.shape[1], b.shape[1], a.data_ptr(), a.stride(0),
                                         b.data_ptr(), b.stride(0), filt_len, c0, sgn, mode_id, out.data_ptr(),
                                         _raw_stream(a.device.index if a.device.index is not None else torch.cuda.current_device()))
        _check(rc)

    def tap_correlate_planes(self, along: int, a: torch.Tensor, b: torch.Tensor, filt_len: int, c0: int, sgn: int, mode_id: int,
                             out: torch.Tensor) -> None:
        """The reduction of :meth:`tap_correlate` on operands in their natural layout ``[batch, rows, columns]`` (unit stride along the
        columns, any batch / row strides: strided views of level buffers need no copy).  ``along`` 1: along the columns
        (``out[t] += sum a[b, r, k] * b_ext[b, r, 2k + c0 + sgn t]``), 0: along the rows (``a[b, k, c] * b_ext[b, 2k + c0 + sgn t, c]``).
        C ABI ``mifwt_tap_correlate_planes``."""
        _require_gpu(a)
        lib = load_library()
        if a.stride(-1) != 1:
            a = a.contiguous()
        if b.stride(-1) != 1:
            b = b.contiguous()
        assert a.dim() == 3 and b.dim() == 3 and a.shape[0] == b.shape[0] and a.dtype == b.dtype and out.dtype == torch.float64
        with torch.cuda.device(a.device):
            rc = lib.mifwt_tap_correlate_planes(_DTYPE_IDS[a.dtype], along, a.shape[0], a.shape[1], a.shape[2], b.shape[1], b.shape[2], a.data_ptr(),
                                                a.stride(0), a.stride(1), b.data_ptr(), b.stride(0), b.stride(1), filt_len, c0, sgn, mode_id,
                                                out.data_ptr(), _raw_stream(a.device.index if a.device.index is not None else torch.cuda.current_device()))
        _check(rc)

    def analysis_outer(self, x: torch.Tensor, dec_lo, dec_hi, mode_id: int):
        """One 1-D analysis level along the MIDDLE axis of ``x`` [B, N, C] (unit stride along C, any batch / row strides) ->
        ``(lo, hi)`` [B, M, C] each (planes of one [B, 2, M, C] buffer): the streaming outer-axis kernel on its own (C ABI
        ``mifwt_dwt1_fwd_outer``); taps as host numbers or :class:`DevTaps`."""
        _require_gpu(x)
        lib = load_library()
        if x.stride(-1) != 1:
            x = x.contiguous()
        B, N, C = x.shape
        flen = len(dec_lo)
        M = (N + flen - 1) // 2
        buf = torch.empty((B, 2, M, C), dtype=x.dtype, device=x.device)
        if B and N and C:
            dev_taps = _is_dev(dec_lo)
            lo = None if dev_taps else _taps_array(dec_lo)
            hi = None if dev_taps else _taps_array(dec_hi)
            with torch.cuda.device(x.device):
                rc = lib.mifwt_dwt1_fwd_outer(_DTYPE_IDS[x.dtype], B, N, C, x.data_ptr(), x.stride(0), x.stride(1), buf.data_ptr(),
                                              buf.data_ptr() + M * C * x.element_size(), 2 * M * C, C, mode_id, flen, lo, hi,
                                              dec_lo.ptr if dev_taps else None, dec_hi.ptr if dev_taps else None,
                                              _raw_stream(x.device.index if x.device.index is not None else torch.cuda.current_device()))
            _check(rc)
        return buf[:, 0], buf[:, 1]

    def synthesis_outer(self, lo: torch.Tensor, hi: torch.Tensor, rec_lo, rec_hi, n_out: int) -> torch.Tensor:
        """One 1-D synthesis level along the MIDDLE axis: ``lo``, ``hi`` [B, M, C] (unit stride along C) -> [B, n_out, C]
        (C ABI ``mifwt_dwt1_inv_outer``); taps as host numbers or :class:`DevTaps`."""
        _require_gpu(lo)
        lib = load_library()
        if lo.stride(-1) != 1:
            lo = lo.contiguous()
        if hi.stride(-1) != 1:
            hi = hi.contiguous()
        B, M, C = lo.shape
        flen = len(rec_lo)
        y = torch.empty((B, n_out, C), dtype=lo.dtype, device=lo.device)
        if B and M and C:
            dev_taps = _is_dev(rec_lo)
            tl = None if dev_taps else _taps_array(rec_lo)
            th = None if dev_taps else _taps_array(rec_hi)
            with torch.cuda.device(lo.device):
                rc = lib.mifwt_dwt1_inv_outer(_DTYPE_IDS[lo.dtype], B, M, n_out, C, lo.data_ptr(), lo.stride(0), lo.stride(1), hi.data_ptr(), hi.stride(0),
                                              hi.stride(1), y.data_ptr(), n_out * C, C, flen, tl, th, rec_lo.ptr if dev_taps else None,
                                              rec_hi.ptr if dev_taps else None,
                                              _raw_stream(lo.device.index if lo.device.index is not None else torch.cuda.current_device()))
            _check(rc)
        return y

    def tap_correlate_dilated(self, a: torch.Tensor, b: torch.Tensor, filt_len: int, c0: int, tstep: int, out: torch.Tensor) -> None:
        """``out[t] += sum_{row, k} a[row, k] * b[row, (k + c0 + tstep t) mod N]`` (C ABI ``mifwt_tap_correlate_dilated``): the tap
        gradients of the stationary levels; ``a``, ``b`` [rows, N] (contiguous samples), ``out`` float64 [filt_len]."""
        _require_gpu(a)
        lib = load_library()
        if a.stride(-1) != 1:
            a = a.contiguous()
        if b.stride(-1) != 1:
            b = b.contiguous()
        assert a.dim() == 2 and a.shape == b.shape and out.dtype == torch.float64
        with torch.cuda.device(a.device):
            rc = lib.mifwt_tap_correlate_dilated(_DTYPE_IDS[a.dtype], a.shape[0], a.shape[1], a.data_ptr(), a.stride(0), b.data_ptr(),
                                                 b.stride(0), filt_len, c0, tstep, out.data_ptr(),
                                                 _raw_stream(a.device.index if a.device.index is not None else torch.cuda.current_device()))
        _check(rc)

    @staticmethod
    def _run(p: _Plan, direction: int, anchor: torch.Tensor, call, kid: Optional[int] = None, dtaps: bool = False) -> None:
        dev = anchor.device
        if dev.index is not None and dev.index != torch.cuda.current_device():
            with torch.cuda.device(dev):
                return HipLevelEngine._run(p, direction, anchor, call, kid, dtaps)
        # (device-resident taps: the fused 2-D kernels where they read device taps, else the generic passes — whose scratch differs
        # from the plan's route; mifwt_kernel_id_dtaps says which)
        wsb = int(_lib.mifwt_workspace_bytes_dtaps(p.ref, direction)) if dtaps else p.ws_bytes
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev) if wsb else None
        if level_events is None:
            rc = call(ws.data_ptr() if ws is not None else None, wsb, _raw_stream(dev.index if dev.index is not None else torch.cuda.current_device()))
        else:
            stream = torch.cuda.current_stream(dev)
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record(stream)
            rc = call(ws.data_ptr() if ws is not None else None, wsb, stream.cuda_stream)
            ev[1].record(stream)
            d = p.desc
            kid_run = (int(_lib.mifwt_kernel_id_dtaps(p.ref, direction)) if hasattr(_lib, "mifwt_kernel_id_dtaps") else 0) if dtaps else (p.kid if kid is None else kid)
            level_events.append((("fwd", "inv", "fwd_adj", "inv_adj")[direction], kid_run, tuple(d.sig_extent[: d.ndim]), ev[0], ev[1]))
        if rc != 0:
            _check(rc)
        # the scratch block returns to the caching allocator when `ws` dies; the allocator only hands it to
        # later work on the SAME stream (stream-ordered reuse), so the level that is still queued keeps it intact


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
