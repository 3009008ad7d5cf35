"""CPU ORACLE for the padded-convolution fast wavelet transform of ptwt (v0lta/PyTorch-Wavelet-Toolbox).

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product package (``pytorch-wavelet-toolbox_amd``)
never does and has no CPU fallback.

It is a plain-numpy restatement of the reference algorithm, written from the reference's behaviour
(file:line citations are to /root/reference, ptwt 1.0.2-dev):

* analysis, one level, one axis  = boundary-pad, then stride-2 cross-correlation with the *flipped*
  decomposition pair (src/ptwt/conv_transform.py:33-66,133-139; flip: src/ptwt/_util.py:863-865);
* pad amounts ``padl = (2L-3)//2``, ``padr = (2L-3)//2 + N%2`` (src/ptwt/_util.py:198-228);
* boundary rules zero / constant / reflect / periodic / symmetric (src/ptwt/_util.py:36-44,163-195 and
  src/ptwt/constants.py:85-108);
* N-D analysis = the outer-product filter bank (src/ptwt/_util.py:870-936), i.e. the 1-D step applied
  along every transformed axis (separable: src/ptwt/separable_conv_transform.py:38-72);
* synthesis, one level, one axis = stride-2 transposed convolution with the un-flipped reconstruction pair,
  then crop ``L-2`` on both sides, one more at the end when the next finer detail band is one shorter
  (src/ptwt/conv_transform.py:184-199, src/ptwt/_util.py:231-244);
* containers: list (1-D), tuple of 3-tuples (2-D), tuple of dicts (3-D and separable)
  (src/ptwt/conv_transform.py:140-143, conv_transform_2.py:145-153, conv_transform_3.py:128-143,
  separable_conv_transform.py:146-153).

Filter taps come from the third-party dependency PyWavelets (unpinned in the reference's pyproject.toml;
this image ships 1.1.1 in /opt/conda).  The taps of all 106 discrete wavelets are frozen in
``tests/golden/pywt_filter_banks.json``.

PARITY PINNING: this oracle is checked (tests/test_oracle.py) against
  (1) the reference's known-answer test (unscaled Haar, tests/test_convolution_fwt.py:98-118),
  (2) golden outputs of the real ``pywt.wavedec/wavedec2/wavedecn`` (tests/golden/pywt_*.npz) — the
      ground truth every hot-path test of the reference uses,
  (3) golden outputs of the reference itself, imported in the build container
      (tests/golden/ptwt_ref_*.npz, made by tests/golden/make_ptwt_ref_goldens.py),
  (4) the live reference when /root/reference is present.
"""
from __future__ import annotations

import json
import math
import os
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

MODES = ("zero", "constant", "reflect", "periodic", "symmetric")

_HERE = os.path.dirname(os.path.abspath(__file__))
_BANK_JSON = os.path.join(_HERE, "..", "tests", "golden", "pywt_filter_banks.json")
_BANKS: Optional[dict] = None

FilterBank = Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]


def filter_bank(wavelet) -> FilterBank:
    """Resolve a wavelet argument to (dec_lo, dec_hi, rec_lo, rec_hi) float64 arrays in pywt order.

    Mirrors the accepted forms of the reference (src/ptwt/_util.py:71-126): name string, object with a
    ``filter_bank`` attribute, or a 4-tuple of sequences.
    """
    global _BANKS
    if isinstance(wavelet, str):
        if _BANKS is None:
            with open(_BANK_JSON) as f:
                _BANKS = json.load(f)
        name = "haar" if wavelet == "db1" and "db1" not in _BANKS else wavelet
        fb = _BANKS[name]
        bank = (fb["dec_lo"], fb["dec_hi"], fb["rec_lo"], fb["rec_hi"])
    elif hasattr(wavelet, "filter_bank"):
        bank = wavelet.filter_bank
    else:
        bank = wavelet
    out = []
    for b in bank:
        if hasattr(b, "detach"):
            b = b.detach().cpu().numpy()
        out.append(np.asarray(b, dtype=np.float64).reshape(-1))
    return tuple(out)  # type: ignore[return-value]


def dwt_max_level(data_len: int, filt_len: int) -> int:
    """pywt.dwt_max_level: floor(log2(N / (L - 1))), 0 when N < L - 1 (reference call site
    src/ptwt/conv_transform.py:129-131)."""
    if data_len < filt_len - 1:
        return 0
    return int(math.floor(math.log2(data_len / (filt_len - 1.0))))


def get_pad(data_len: int, filt_len: int) -> Tuple[int, int]:
    """(padl, padr) of one axis (src/ptwt/_util.py:198-228)."""
    padr = (2 * filt_len - 3) // 2
    padl = (2 * filt_len - 3) // 2
    padr += data_len % 2
    return padl, padr


def ext_index(i: np.ndarray, n: int, mode: str) -> np.ndarray:
    """Map extended-signal indices to source indices; -1 marks an implicit zero.

    zero/constant/reflect/periodic are torch's constant/replicate/reflect/circular pads
    (src/ptwt/_util.py:36-44); symmetric is the half-sample mirror applied repeatedly
    (src/ptwt/_util.py:163-176).
    """
    i = np.asarray(i, dtype=np.int64)
    inside = (i >= 0) & (i < n)
    if mode == "zero":
        return np.where(inside, i, -1)
    if mode == "constant":
        return np.clip(i, 0, n - 1)
    if mode == "periodic":
        return np.mod(i, n)
    if mode == "symmetric":
        p = np.mod(i, 2 * n)
        return np.where(p < n, p, 2 * n - 1 - p)
    if mode == "reflect":
        if n == 1:
            return np.zeros_like(i)
        p = np.mod(i, 2 * (n - 1))
        return np.where(p < n, p, 2 * (n - 1) - p)
    raise ValueError(f"Padding mode not supported: {mode}")


def check_pad_like_torch(n: int, padl: int, padr: int, mode: str) -> None:
    """torch.nn.functional.pad refuses reflect pad >= N and circular pad > N; the reference lets that
    RuntimeError surface."""
    if mode == "reflect" and (padl >= n or padr >= n):
        raise RuntimeError("reflect padding must be smaller than the input extent")
    if mode == "periodic" and (padl > n or padr > n):
        raise RuntimeError("circular padding must not exceed the input extent")


def pad_axis(x: np.ndarray, padl: int, padr: int, mode: str, axis: int) -> np.ndarray:
    n = x.shape[axis]
    check_pad_like_torch(n, padl, padr, mode)
    idx = ext_index(np.arange(-padl, n + padr), n, mode)
    out = np.take(x, np.where(idx < 0, 0, idx), axis=axis)
    if mode == "zero":
        shape = [1] * x.ndim
        shape[axis] = -1
        out = out * (idx >= 0).astype(x.dtype).reshape(shape)
    return out


def dwt_axis(x: np.ndarray, dec_lo, dec_hi, mode: str, axis: int = -1) -> Tuple[np.ndarray, np.ndarray]:
    """One analysis level along one axis: returns (approx, detail)."""
    lo = np.asarray(dec_lo, dtype=x.dtype)
    hi = np.asarray(dec_hi, dtype=x.dtype)
    flen = lo.shape[0]
    n = x.shape[axis]
    padl, padr = get_pad(n, flen)
    xp = np.moveaxis(pad_axis(x, padl, padr, mode, axis), axis, -1)
    m = (xp.shape[-1] - flen) // 2 + 1
    a = np.zeros(xp.shape[:-1] + (m,), dtype=x.dtype)
    d = np.zeros_like(a)
    # cross-correlation with the flipped taps, stride 2:  c[k] = sum_j flip(h)[j] * xp[2k + j]
    for j in range(flen):
        seg = xp[..., j : j + 2 * (m - 1) + 1 : 2]
        a += lo[flen - 1 - j] * seg
        d += hi[flen - 1 - j] * seg
    return np.moveaxis(a, -1, axis), np.moveaxis(d, -1, axis)


def idwt_axis(a: np.ndarray, d: np.ndarray, rec_lo, rec_hi, axis: int = -1, trim_end: int = 0) -> np.ndarray:
    """One synthesis level along one axis, cropped like the reference (L-2 each side, + trim_end)."""
    lo = np.asarray(rec_lo, dtype=a.dtype)
    hi = np.asarray(rec_hi, dtype=a.dtype)
    flen = lo.shape[0]
    a = np.moveaxis(a, axis, -1)
    d = np.moveaxis(d, axis, -1)
    m = a.shape[-1]
    full = np.zeros(a.shape[:-1] + (2 * (m - 1) + flen,), dtype=a.dtype)
    # transposed convolution, stride 2:  u[2k + t] += a[k] * g_lo[t] + d[k] * g_hi[t]
    for t in range(flen):
        full[..., t : t + 2 * (m - 1) + 1 : 2] += lo[t] * a + hi[t] * d
    pad = (2 * flen - 3) // 2
    end = full.shape[-1] - pad - trim_end
    return np.moveaxis(full[..., pad:end], -1, axis)


def adjust_trim(res_size: int, next_size: int) -> int:
    """src/ptwt/_util.py:231-244 expressed on the already-cropped size."""
    if next_size == res_size:
        return 0
    if next_size == res_size - 1:
        return 1
    raise AssertionError("padding error, please check if dec and rec wavelets are identical.")


# --------------------------------------------------------------------------------------- axes plumbing
def _norm_axes(axes, ndim_data: int, n: int) -> Tuple[int, ...]:
    if axes is None:
        axes = tuple(range(-n, 0))
    if isinstance(axes, int):
        axes = (axes,)
    axes = tuple(a + ndim_data if a < 0 else a for a in axes)
    if len(axes) != n or len(set(axes)) != n:
        raise ValueError("bad axes")
    return axes


def _check(x: np.ndarray, n: int) -> None:
    if x.dtype not in (np.float32, np.float64):
        raise ValueError(f"Input dtype {x.dtype} not supported")
    if x.ndim < n:
        raise ValueError(f"At least {n} input dimensions required.")


def _dwtn(x: np.ndarray, bank: FilterBank, mode: str, axes: Sequence[int]) -> Dict[str, np.ndarray]:
    """Single-level N-D analysis; key char i <-> axes[i] ('a' low-pass, 'd' high-pass)."""
    out = {"": x}
    for ax in axes:
        nxt = {}
        for key, val in out.items():
            lo, hi = dwt_axis(val, bank[0], bank[1], mode, ax)
            nxt[key + "a"] = lo
            nxt[key + "d"] = hi
        out = nxt
    return out


def _idwtn(bands: Dict[str, np.ndarray], bank: FilterBank, axes: Sequence[int], trims: Sequence[int]) -> np.ndarray:
    cur = dict(bands)
    for pos in reversed(range(len(axes))):
        nxt = {}
        for key in sorted({k[:pos] for k in cur}):
            nxt[key] = idwt_axis(cur[key + "a"], cur[key + "d"], bank[2], bank[3], axes[pos], trims[pos])
        cur = nxt
    return cur[""]


def _max_level(shape: Sequence[int], flen: int) -> int:
    return min(dwt_max_level(n, flen) for n in shape)


# --------------------------------------------------------------------------------------- public mirror
def wavedec(x, wavelet, *, mode="reflect", level=None, axis=-1) -> List[np.ndarray]:
    x = np.asarray(x)
    _check(x, 1)
    if mode not in MODES:
        raise ValueError(f"Padding mode not supported: {mode}")
    bank = filter_bank(wavelet)
    (ax,) = _norm_axes(axis, x.ndim, 1)
    if level is None:
        level = dwt_max_level(x.shape[ax], len(bank[0]))
    out, cur = [], x
    for _ in range(level):
        cur, det = dwt_axis(cur, bank[0], bank[1], mode, ax)
        out.append(det)
    out.append(cur)
    return out[::-1]


def waverec(coeffs, wavelet, *, axis=-1) -> np.ndarray:
    coeffs = [np.asarray(c) for c in coeffs]
    bank = filter_bank(wavelet)
    (ax,) = _norm_axes(axis, coeffs[0].ndim, 1)
    flen = len(bank[2])
    cur = coeffs[0]
    for pos, det in enumerate(coeffs[1:]):
        trim = 0
        if pos + 2 < len(coeffs):
            trim = adjust_trim(2 * cur.shape[ax] - flen + 2, coeffs[pos + 2].shape[ax])
        cur = idwt_axis(cur, det, bank[2], bank[3], ax, trim)
    return cur


def _wavedecn(x, wavelet, mode, level, axes, n):
    x = np.asarray(x)
    _check(x, n)
    if mode not in MODES:
        raise ValueError(f"Padding mode not supported: {mode}")
    bank = filter_bank(wavelet)
    axes = _norm_axes(axes, x.ndim, n)
    if level is None:
        level = _max_level([x.shape[a] for a in axes], len(bank[0]))
    details, cur = [], x
    for _ in range(level):
        bands = _dwtn(cur, bank, mode, axes)
        cur = bands.pop("a" * n)
        details.append(bands)
    return cur, details[::-1]


def _waverecn(approx, details, wavelet, axes, n, trim_inputs=False):
    bank = filter_bank(wavelet)
    cur = np.asarray(approx)
    axes = _norm_axes(axes, cur.ndim, n)
    flen = len(bank[2])
    for pos, det in enumerate(details):
        det = {k: np.asarray(v) for k, v in det.items()}
        any_det = next(iter(det.values()))
        if trim_inputs:
            # the separable reference crops the running approximation to the detail shape
            # (src/ptwt/separable_conv_transform.py:94-97) instead of cropping the synthesis output
            cur = cur[tuple(slice(0, s) for s in any_det.shape)]
            trims = [0] * n
        else:
            for v in det.values():
                if v.shape != cur.shape:
                    raise ValueError("All coefficients on each level must have the same shape")
            trims = [0] * n
            if pos + 1 < len(details):
                nxt = next(iter(details[pos + 1].values()))
                trims = [adjust_trim(2 * cur.shape[a] - flen + 2, np.asarray(nxt).shape[a]) for a in axes]
        bands = dict(det)
        bands["a" * n] = cur
        cur = _idwtn(bands, bank, axes, trims)
    return cur


def wavedec2(x, wavelet, *, mode="reflect", level=None, axes=(-2, -1)):
    cur, details = _wavedecn(x, wavelet, mode, level, axes, 2)
    # H = hi on axes[0] & lo on axes[1] ('da'), V = 'ad', D = 'dd'  (src/ptwt/_util.py:901-905)
    return (cur,) + tuple((d["da"], d["ad"], d["dd"]) for d in details)


def waverec2(coeffs, wavelet, *, axes=(-2, -1)):
    details = []
    for c in coeffs[1:]:
        if not isinstance(c, tuple) or len(c) != 3:
            raise ValueError("Unexpected detail coefficient type")
        details.append({"da": c[0], "ad": c[1], "dd": c[2]})
    return _waverecn(coeffs[0], details, wavelet, axes, 2)


def wavedec3(x, wavelet, *, mode="zero", level=None, axes=(-3, -2, -1)):
    cur, details = _wavedecn(x, wavelet, mode, level, axes, 3)
    order = ("aad", "ada", "add", "daa", "dad", "dda", "ddd")
    return (cur,) + tuple({k: d[k] for k in order} for d in details)


def waverec3(coeffs, wavelet, *, axes=(-3, -2, -1)):
    for c in coeffs[1:]:
        if not isinstance(c, dict) or len(c) != 7:
            raise ValueError("Unexpected detail coefficient type")
    return _waverecn(coeffs[0], list(coeffs[1:]), wavelet, axes, 3)


def _fs_order(n: int) -> List[str]:
    """Key insertion order of the separable reference's recursion
    (src/ptwt/separable_conv_transform.py:63-72: last axis first, new char prepended)."""
    # the recursion visits the 'a'+key subtree completely before 'd'+key: depth-first order
    def rec(key: str, out: List[str]):
        if len(key) == n:
            out.append(key)
            return
        rec("a" + key, out)
        rec("d" + key, out)

    out: List[str] = []
    rec("", out)
    return out


def fswavedec(x, wavelet, n, *, mode="reflect", level=None, axes=None):
    cur, details = _wavedecn(x, wavelet, mode, level, axes, n)
    order = [k for k in _fs_order(n) if k != "a" * n]
    return (cur,) + tuple({k: d[k] for k in order} for d in details)


def fswaverec(coeffs, wavelet, n, *, axes=None):
    if not isinstance(coeffs[0], np.ndarray):
        raise ValueError("approximation tensor must be first in coefficient list.")
    if not all(isinstance(c, dict) for c in coeffs[1:]):
        raise ValueError("All entries after approximation tensor must be dicts.")
    return _waverecn(coeffs[0], list(coeffs[1:]), wavelet, axes, n, trim_inputs=True)


def fswavedec2(x, wavelet, **kw):
    return fswavedec(x, wavelet, 2, **kw)


def fswavedec3(x, wavelet, **kw):
    return fswavedec(x, wavelet, 3, **kw)


def fswaverec2(coeffs, wavelet, **kw):
    return fswaverec(coeffs, wavelet, 2, **kw)


def fswaverec3(coeffs, wavelet, **kw):
    return fswaverec(coeffs, wavelet, 3, **kw)
