"""CPU PORT of the reference's wavedec2 data path, used ONLY as the ``cpu_baseline`` leg of bench.py and in
tests (test infrastructure — never imported by the product package).

It issues the same ATen op sequence per level as ptwt on a CPU tensor — boundary pad, then one dense
stride-2 ``conv2d`` with the ``[4,1,L,L]`` outer-product filter bank, then channel split
(reference src/ptwt/conv_transform_2.py:142-149; filters src/ptwt/_util.py:886-907; pad amounts
src/ptwt/_util.py:198-228) — so its wall time on the host cores is the reference's CPU path minus Python
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
