"""1-D padded-convolution FWT: ``wavedec`` / ``waverec`` (API of reference src/ptwt/conv_transform.py:69-204)."""
from __future__ import annotations

from typing import List, Optional, Sequence, Union

import torch

from . import _fwt
from .constants import BoundaryMode, Wavelet

__all__ = ["wavedec", "waverec"]


def wavedec(data: torch.Tensor, wavelet: Union[Wavelet, str], *, mode: BoundaryMode = "reflect",
            level: Optional[int] = None, axis: int = -1) -> List[torch.Tensor]:
    """Multi-level 1-D analysis along ``axis``; returns ``[cA_n, cD_n, ..., cD_1]``.

    Drop-in for ``ptwt.wavedec`` (src/ptwt/conv_transform.py:69-143): same arguments, defaults, container,
    coefficient lengths ``floor((N + L - 1) / 2)`` per level and error types.  Every level is one fused HIP
    kernel (boundary extension by index map + both filters), never a padded copy.
    """
    layout, approx, bufs = _fwt.analysis(data, wavelet, mode, level, axis, 1)
    return _fwt.pack_1d(layout, approx, bufs)


def waverec(coeffs: Sequence[torch.Tensor], wavelet: Union[Wavelet, str], *, axis: _fwt.AxisHint = None) -> torch.Tensor:
    """Inverse of :func:`wavedec` (src/ptwt/conv_transform.py:146-204); odd-length inputs come back one
    sample longer, exactly as in the reference."""
    if not isinstance(coeffs, list):
        coeffs = list(coeffs)
    if not coeffs or not isinstance(coeffs[0], torch.Tensor):
        raise ValueError("First element of coeffs must be the approximation coefficient tensor.")
    return _fwt.synthesis(coeffs[0], [[c] for c in coeffs[1:]], wavelet, axis, 1, separable=False)
