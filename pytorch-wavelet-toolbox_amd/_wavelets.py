"""Built-in discrete wavelet filter banks, so the engine does not need PyWavelets at run time.

The reference obtains taps with ``pywt.Wavelet(name).filter_bank`` (src/ptwt/_util.py:71-126) and the
default level with ``pywt.dwt_max_level`` / ``dwtn_max_level`` (src/ptwt/conv_transform.py:129-131,
conv_transform_2.py:137-138).  ``_filter_banks.json`` holds the taps of all 106 discrete wavelets of
PyWavelets 1.1.1; any object exposing ``filter_bank`` (a ``pywt.Wavelet``, a learnable-wavelet module) or a
4-tuple of sequences/tensors is accepted as well, exactly as in the reference.
"""
from __future__ import annotations

import json
import math
import os
from functools import lru_cache
from typing import List, Sequence, Tuple

_TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_filter_banks.json")
_ALIASES = {"db1": "haar"}


@lru_cache(maxsize=1)
def _table() -> dict:
    with open(_TABLE) as f:
        return json.load(f)


def wavelist() -> List[str]:
    """Names of the built-in discrete wavelets."""
    return sorted(k for k in _table() if not k.startswith("_"))


class BuiltinWavelet:
    """Minimal ``pywt.Wavelet`` look-alike backed by the built-in table."""

    def __init__(self, name: str):
        table = _table()
        key = name if name in table else _ALIASES.get(name, name)
        if key not in table or key.startswith("_"):
            raise ValueError(f"Unknown wavelet name '{name}', check wavelist() for the list of available builtin wavelets.")
        self.name = name
        self.dec_lo, self.dec_hi, self.rec_lo, self.rec_hi = (list(t) for t in table[key])
        self.dec_len = len(self.dec_lo)
        self.rec_len = len(self.rec_lo)

    @property
    def filter_bank(self) -> Tuple[List[float], List[float], List[float], List[float]]:
        return self.dec_lo, self.dec_hi, self.rec_lo, self.rec_hi

    def __len__(self) -> int:
        return self.dec_len

    def __repr__(self) -> str:
        return f"BuiltinWavelet({self.name!r}, len={self.dec_len})"


@lru_cache(maxsize=256)
def _named(name: str) -> BuiltinWavelet:
    return BuiltinWavelet(name)


def as_wavelet(wavelet):
    """str -> built-in wavelet object; anything else is returned unchanged (src/ptwt/_util.py:71-84)."""
    return _named(wavelet) if isinstance(wavelet, str) else wavelet


def _to_floats(seq) -> Tuple[float, ...]:
    """Host copy of a tap sequence.  The kernels take the taps by value in their launch arguments, so a tensor-valued filter
    (a learnable parameter on the GPU) is read back on EVERY top-level transform call, like the reference, which always reads the
    live tensor (src/ptwt/_util.py:115-132).  Nothing is cached across calls: an in-place update through ``p.data`` (hand-written
    SGD, weight clipping) does not bump ``p._version``, so no cheap validity check exists."""
    if hasattr(seq, "detach"):
        return tuple(float(v) for v in seq.detach().reshape(-1).cpu().tolist())
    return tuple(float(v) for v in seq)


def _bank_to_floats(bank):
    """The four filters of a bank as host floats; tensor-valued banks that live on one device travel in ONE device-to-host copy."""
    if all(hasattr(t, "detach") for t in bank) and len({(t.device, t.dtype) for t in bank}) == 1 and bank[0].is_cuda:
        import torch

        flat = torch.cat([t.detach().reshape(-1) for t in bank]).cpu().tolist()
        out, pos = [], 0
        for t in bank:
            n = t.numel()
            out.append(tuple(float(v) for v in flat[pos:pos + n]))
            pos += n
        return tuple(out)
    return tuple(_to_floats(t) for t in bank)


@lru_cache(maxsize=256)
def _named_taps(name: str):
    return tuple(tuple(t) for t in _named(name).filter_bank)


def host_taps(wavelet) -> Tuple[Tuple[float, ...], Tuple[float, ...], Tuple[float, ...], Tuple[float, ...]]:
    """(dec_lo, dec_hi, rec_lo, rec_hi) as host floats in PyWavelets order (NOT flipped; the kernels index
    the taps so that the flip of src/ptwt/_util.py:863-865 is implicit).

    The kernels take the taps by value in their launch arguments, so tensor-valued filter banks are read
    back to the host here (one small D2H copy when they live on the GPU).
    """
    if isinstance(wavelet, str):
        return _named_taps(wavelet)
    bank = wavelet if isinstance(wavelet, tuple) else wavelet.filter_bank
    if len(bank) != 4:
        raise ValueError("a filter bank must hold (dec_lo, dec_hi, rec_lo, rec_hi)")
    taps = _bank_to_floats(bank)
    if not (len(taps[0]) == len(taps[1]) and len(taps[2]) == len(taps[3])):
        raise ValueError("low- and high-pass filters must have the same length")
    return taps  # type: ignore[return-value]


# ---- device-resident taps -------------------------------------------------------------------------------------------------------
# A filter bank given as four tensors on the GPU (a learnable wavelet, src/ptwt/_util.py:115-132) can stay there: the level kernels
# read their taps from device memory (C ABI mifwt_*_dtaps), so a call copies nothing to the host, synchronises nothing and can be
# captured into a HIP graph.  Since round 6 that costs nothing on the planes a training step runs on: the fused per-level kernels (LDS
# tiles, one level of the streaming kernels, the border kernels, the streaming axis passes) read device taps, bit-identical to host
# taps; only what they do not serve (3-D levels, matrix-core and 16-tap streaming kernels) falls to the generic per-axis passes.  The
# multi-level launches are not used with device taps (a call that is not learnable reads its bank to the host once and keeps them).
#   "auto"   (default) device taps for banks that are learnable (a tensor requires grad, grad mode on) or while a HIP graph is being
#            captured; anything else is read to the host once per call (fused kernels incl. the multi-level launches);
#   "always" every tensor-valued bank on the GPU stays there;   "never" round 4's behaviour (one device-to-host copy per call).
_device_taps_mode = "auto"


def set_device_taps(mode: str) -> None:
    """"auto" | "always" | "never": when a tensor-valued filter bank on the GPU is handed to the kernels as device memory."""
    global _device_taps_mode
    if mode not in ("auto", "always", "never"):
        raise ValueError("mode must be 'auto', 'always' or 'never'")
    _device_taps_mode = mode


def device_bank(wavelet, device):
    """The four filters as tensors if this call keeps its taps on the GPU (see above), else None."""
    if _device_taps_mode == "never" or isinstance(wavelet, str):
        return None
    import torch

    bank = wavelet if isinstance(wavelet, tuple) else getattr(wavelet, "filter_bank", ())
    if len(bank) != 4 or not all(isinstance(t, torch.Tensor) and t.is_cuda and t.device == device for t in bank):
        return None
    if not (bank[0].numel() == bank[1].numel() and bank[2].numel() == bank[3].numel()):
        raise ValueError("low- and high-pass filters must have the same length")
    if _device_taps_mode == "always":
        return tuple(bank)
    learnable = torch.is_grad_enabled() and any(t.requires_grad for t in bank)
    if learnable or torch.cuda.is_current_stream_capturing():
        return tuple(bank)
    return None


def filter_length(wavelet) -> int:
    """Number of taps for any accepted wavelet form (src/ptwt/_util.py:87-92)."""
    if isinstance(wavelet, tuple):
        return int(wavelet[0].shape[0]) if hasattr(wavelet[0], "shape") else len(wavelet[0])
    return len(as_wavelet(wavelet))


def dwt_max_level(data_len: int, filt_len: int) -> int:
    """``pywt.dwt_max_level``: floor(log2(N / (L - 1))), 0 when the signal is shorter than L - 1."""
    if filt_len < 2:
        raise ValueError("filter length must be at least 2")
    if data_len < filt_len - 1:
        return 0
    return int(math.floor(math.log2(data_len // (filt_len - 1))))


def dwtn_max_level(shape: Sequence[int], filt_len: int) -> int:
    return min(dwt_max_level(int(n), filt_len) for n in shape)
