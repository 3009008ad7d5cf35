"""2-D padded-convolution FWT: ``wavedec2`` / ``waverec2`` (API of reference src/ptwt/conv_transform_2.py:74-253)."""
from __future__ import annotations

from typing import Optional, Tuple, Union

import torch

from . import _fwt
from .constants import BoundaryMode, Wavelet, WaveletCoeff2d

__all__ = ["wavedec2", "waverec2"]


def wavedec2(data: torch.Tensor, wavelet: Union[Wavelet, str], *, mode: BoundaryMode = "reflect",
             level: Optional[int] = None, axes: Tuple[int, int] = (-2, -1)) -> WaveletCoeff2d:
    """Multi-level 2-D analysis; returns ``(cA_n, (H,V,D)_n, ..., (H,V,D)_1)``.

    Drop-in for ``ptwt.wavedec2`` (src/ptwt/conv_transform_2.py:74-157).  The reference's dense
    ``[4,1,L,L]`` conv2d is an outer product of the 1-D pair (src/ptwt/_util.py:886-907); the engine
    computes the same four bands separably in one fused kernel per level.
    """
    layout, approx, bufs = _fwt.analysis(data, wavelet, mode, level, axes, 2)
    return _fwt.pack_2d(layout, approx, bufs)


def waverec2(coeffs: WaveletCoeff2d, wavelet: Union[Wavelet, str], *, axes: _fwt.AxisHint = None) -> torch.Tensor:
    """Inverse of :func:`wavedec2` (src/ptwt/conv_transform_2.py:160-253)."""
    if len(coeffs) == 0 or not isinstance(coeffs[0], torch.Tensor):
        raise ValueError("First element of coeffs must be the approximation coefficient tensor.")
    levels = []
    for c in coeffs[1:]:
        if not isinstance(c, tuple) or len(c) != 3:
            raise ValueError(
                f"Unexpected detail coefficient type: {type(c)}. Detail coefficients must be a 3-tuple of "
                "tensors as returned by wavedec2."
            )
        levels.append([c[1], c[0], c[2]])  # bands 1,2,3 = 'ad' (V), 'da' (H), 'dd' (D)
    return _fwt.synthesis(coeffs[0], levels, wavelet, axes, 2, separable=False)
