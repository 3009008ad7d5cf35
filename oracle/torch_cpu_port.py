"""CPU PORT of the reference's data paths, used ONLY as the ``cpu_baseline`` leg of bench.py and in
tests (test infrastructure — never imported by the product package).

It issues the same ATen op sequence per level as ptwt on a CPU tensor — analysis: boundary pad, then one dense
stride-2 ``conv{1,2,3}d`` with the outer-product filter bank (C_in = 1), then channel split
(reference src/ptwt/conv_transform.py:135-139, conv_transform_2.py:142-149, conv_transform_3.py:121-141; filters
src/ptwt/_util.py:886-936; pad amounts src/ptwt/_util.py:198-228); synthesis: ``torch.stack`` + stride-2
``conv_transpose2d`` + crops (conv_transform_2.py:222-249); separable: one 1-D level per axis over the folded tensor
(separable_conv_transform.py:36-72) — so its wall time on the host cores is the reference's CPU path minus Python
glue (<0.1 % of the time per SURVEY.md §3.1).  /root/reference itself cannot travel to the GPU box.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import fwt_oracle as O

_TORCH_MODE = {"zero": "constant", "constant": "replicate", "reflect": "reflect", "periodic": "circular"}


def _pad2(x: torch.Tensor, flen: int, mode: str) -> torch.Tensor:
    padl = (2 * flen - 3) // 2
    pads = (padl, padl + x.shape[-1] % 2, padl, padl + x.shape[-2] % 2)
    if mode == "symmetric":  # torch has no half-sample mirror: index-gather (single fold is enough here)
        for axis, (a, b) in ((-1, pads[:2]), (-2, pads[2:])):
            n = x.shape[axis]
            idx = torch.from_numpy(O.ext_index(range(-a, n + b), n, "symmetric"))
            x = x.index_select(axis, idx)
        return x
    return F.pad(x, pads, mode=_TORCH_MODE[mode])


def wavedec2(x: torch.Tensor, wavelet, *, mode: str = "reflect", level: int = 1):
    """``x``: [B, H, W] CPU tensor.  Returns ``(cA, (H, V, D)_n, ..., (H, V, D)_1)`` like ptwt."""
    dec_lo, dec_hi, _, _ = O.filter_bank(wavelet)
    lo = torch.tensor(dec_lo[::-1].copy(), dtype=x.dtype)
    hi = torch.tensor(dec_hi[::-1].copy(), dtype=x.dtype)
    bank = torch.stack([torch.outer(lo, lo), torch.outer(hi, lo), torch.outer(lo, hi), torch.outer(hi, hi)]).unsqueeze(1)
    cur = x.unsqueeze(1)
    out = []
    for _ in range(level):
        res = F.conv2d(_pad2(cur, len(dec_lo), mode), bank, stride=2)
        cur, lh, hl, hh = torch.split(res, 1, 1)
        out.append((lh.squeeze(1), hl.squeeze(1), hh.squeeze(1)))
    return (cur.squeeze(1), *out[::-1])


def _pad_axes(x: torch.Tensor, flen: int, mode: str, naxes: int) -> torch.Tensor:
    """Boundary extension of the last ``naxes`` axes of [B, 1, ...] (src/ptwt/_util.py:198-228)."""
    padl = (2 * flen - 3) // 2
    if mode == "symmetric":
        for axis in range(-naxes, 0):
            n = x.shape[axis]
            idx = torch.from_numpy(O.ext_index(range(-padl, n + padl + n % 2), n, "symmetric"))
            x = x.index_select(axis, idx)
        return x
    pads = []
    for axis in range(-1, -naxes - 1, -1):
        pads += [padl, padl + x.shape[axis] % 2]
    return F.pad(x, pads, mode=_TORCH_MODE[mode])


def _flipped(wavelet, dtype, which=(0, 1)):
    bank = O.filter_bank(wavelet)
    return tuple(torch.tensor(bank[i][::-1].copy(), dtype=dtype) for i in which), len(bank[0])


def wavedec(x: torch.Tensor, wavelet, *, mode: str = "reflect", level: int = 1):
    """``x``: [B, N] CPU tensor -> ``[cA_n, cD_n, ..., cD_1]`` (src/ptwt/conv_transform.py:69-143)."""
    (lo, hi), flen = _flipped(wavelet, x.dtype)
    bank = torch.stack([lo, hi]).unsqueeze(1)
    cur = x.unsqueeze(1)
    out = []
    for _ in range(level):
        res = F.conv1d(_pad_axes(cur, flen, mode, 1), bank, stride=2)
        cur, d = torch.split(res, 1, 1)
        out.append(d.squeeze(1))
    return [cur.squeeze(1), *out[::-1]]


def wavedec3(x: torch.Tensor, wavelet, *, mode: str = "zero", level: int = 1):
    """``x``: [B, D, H, W] CPU tensor -> ``(cA, {aad: .., ...}_n, ...)`` (src/ptwt/conv_transform_3.py:76-145)."""
    (lo, hi), flen = _flipped(wavelet, x.dtype)
    filt, keys = [], []
    for a, fa in (("a", lo), ("d", hi)):
        for b, fb in (("a", lo), ("d", hi)):
            for c, fc in (("a", lo), ("d", hi)):
                filt.append(fa[:, None, None] * fb[None, :, None] * fc[None, None, :])
                keys.append(a + b + c)
    bank = torch.stack(filt).unsqueeze(1)
    cur = x.unsqueeze(1)
    out = []
    for _ in range(level):
        res = F.conv3d(_pad_axes(cur, flen, mode, 3), bank, stride=2)
        parts = torch.split(res, 1, 1)
        cur = parts[0]
        out.append({k: v.squeeze(1) for k, v in zip(keys[1:], parts[1:])})
    return (cur.squeeze(1), *out[::-1])


def waverec2(coeffs, wavelet):
    """``(cA, (H, V, D)_n, ..., (H, V, D)_1)`` of [B, h, w] CPU tensors -> [B, H, W] (src/ptwt/conv_transform_2.py:160-253)."""
    (lo, hi), flen = _flipped(wavelet, coeffs[0].dtype, which=(2, 3))
    lo, hi = lo.flip(0), hi.flip(0)  # conv_transpose correlates with the un-flipped reconstruction filters
    bank = torch.stack([torch.outer(lo, lo), torch.outer(hi, lo), torch.outer(lo, hi), torch.outer(hi, hi)]).unsqueeze(1)
    cur = coeffs[0]
    pad = (2 * flen - 3) // 2
    for pos, (h, v, d) in enumerate(coeffs[1:]):
        res = F.conv_transpose2d(torch.stack([cur, h, v, d], 1), bank, stride=2).squeeze(1)
        pb = pr = pad
        if pos + 2 < len(coeffs):
            nxt = coeffs[pos + 2][0].shape
            pb += O.adjust_trim(res.shape[-2] - 2 * pad, nxt[-2])
            pr += O.adjust_trim(res.shape[-1] - 2 * pad, nxt[-1])
        cur = res[..., pad:res.shape[-2] - pb, pad:res.shape[-1] - pr]
    return cur


def waverec(coeffs, wavelet):
    """``[cA_n, cD_n, ..., cD_1]`` of [B, n] CPU tensors -> [B, N] (src/ptwt/conv_transform.py:146-204)."""
    bank = O.filter_bank(wavelet)
    dt = coeffs[0].dtype
    filt = torch.stack([torch.tensor(bank[2].copy(), dtype=dt), torch.tensor(bank[3].copy(), dtype=dt)]).unsqueeze(1)
    flen = len(bank[2])
    pad = (2 * flen - 3) // 2
    cur = coeffs[0]
    for pos, d in enumerate(coeffs[1:]):
        res = F.conv_transpose1d(torch.stack([cur, d], 1), filt, stride=2).squeeze(1)
        pr = pad
        if pos + 2 < len(coeffs):
            pr += O.adjust_trim(res.shape[-1] - 2 * pad, coeffs[pos + 2].shape[-1])
        cur = res[..., pad:res.shape[-1] - pr]
    return cur


def waverec3(coeffs, wavelet):
    """``(cA, {aad: .., ...}_n, ...)`` of [B, d, h, w] CPU tensors -> [B, D, H, W] (src/ptwt/conv_transform_3.py:148-251)."""
    bank = O.filter_bank(wavelet)
    dt = coeffs[0].dtype
    lo, hi = torch.tensor(bank[2].copy(), dtype=dt), torch.tensor(bank[3].copy(), dtype=dt)
    filt, keys = [], []
    for a, fa in (("a", lo), ("d", hi)):
        for b, fb in (("a", lo), ("d", hi)):
            for c, fc in (("a", lo), ("d", hi)):
                filt.append(fa[:, None, None] * fb[None, :, None] * fc[None, None, :])
                keys.append(a + b + c)
    bankt = torch.stack(filt).unsqueeze(1)
    flen = len(bank[2])
    pad = (2 * flen - 3) // 2
    cur = coeffs[0]
    for pos, det in enumerate(coeffs[1:]):
        res = F.conv_transpose3d(torch.stack([cur] + [det[k] for k in keys[1:]], 1), bankt, stride=2).squeeze(1)
        trims = [0, 0, 0]
        if pos + 2 < len(coeffs):
            nxt = next(iter(coeffs[pos + 2].values())).shape
            trims = [O.adjust_trim(res.shape[-3 + a] - 2 * pad, nxt[-3 + a]) for a in range(3)]
        cur = res[..., pad:res.shape[-3] - pad - trims[0], pad:res.shape[-2] - pad - trims[1], pad:res.shape[-1] - pad - trims[2]]
    return cur


def fswavedec2(x: torch.Tensor, wavelet, *, mode: str = "reflect", level: int = 1):
    """``x``: [B, H, W] CPU tensor -> ``(cA, {ad, da, dd}_n, ...)``: one 1-D level along the last axis, then one along the rows of
    both halves, per level (src/ptwt/separable_conv_transform.py:36-72, 187-231)."""
    def level1(t, axis):  # the reference swaps the axis last, folds everything else into the batch and calls wavedec(level=1)
        tt = t.transpose(-1, axis)
        a, d = wavedec(tt.reshape(-1, tt.shape[-1]), wavelet, mode=mode, level=1)
        return tuple(c.reshape(*tt.shape[:-1], c.shape[-1]).transpose(-1, axis) for c in (a, d))
    cur, out = x, []
    for _ in range(level):
        a, d = level1(cur, -1)
        aa, da = level1(a, -2)
        ad, dd = level1(d, -2)
        cur = aa
        out.append({"ad": ad, "da": da, "dd": dd})
    return (cur, *out[::-1])


def fswaverec2(coeffs, wavelet):
    """``(cA, {ad, da, dd}_n, ...)`` of [B, h, w] CPU tensors -> [B, H, W]: per level one 1-D synthesis level along the rows' axis
    for (aa, da) and (ad, dd) — the approximation first cut to the details' shape — and one along the last axis for the two results,
    each over the folded tensor (src/ptwt/separable_conv_transform.py:75-110, 281-313)."""
    def level1(a, d, axis):
        a = a[tuple(slice(0, n) for n in d.shape)]
        ta, td = a.transpose(-1, axis), d.transpose(-1, axis)
        rec = waverec([ta.reshape(-1, ta.shape[-1]), td.reshape(-1, td.shape[-1])], wavelet)
        return rec.reshape(*ta.shape[:-1], rec.shape[-1]).transpose(axis, -1)
    cur = coeffs[0]
    for det in coeffs[1:]:
        cur = level1(level1(cur, det["da"], -2), level1(det["ad"], det["dd"], -2), -1)
    return cur
