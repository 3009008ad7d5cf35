"""3-D padded-convolution FWT: ``wavedec3`` / ``waverec3`` (API of reference src/ptwt/conv_transform_3.py:76-251)."""
from __future__ import annotations

from typing import Optional, Tuple, Union

import torch

from . import _fwt
from .constants import BoundaryMode, Wavelet, WaveletCoeffNd

__all__ = ["wavedec3", "waverec3"]


def wavedec3(data: torch.Tensor, wavelet: Union[Wavelet, str], *, mode: BoundaryMode = "zero",
             level: Optional[int] = None, axes: Tuple[int, int, int] = (-3, -2, -1)) -> WaveletCoeffNd:
    """Multi-level 3-D analysis; returns ``(cA_n, {"aad",...,"ddd"}_n, ..., {...}_1)``.

    Drop-in for ``ptwt.wavedec3`` (src/ptwt/conv_transform_3.py:76-145); note the default mode is
    ``"zero"`` there and here.
    """
    layout, approx, bufs = _fwt.analysis(data, wavelet, mode, level, axes, 3)
    return _fwt.pack_dict(layout, approx, bufs, _fwt._KEYS_ND[3])


def waverec3(coeffs: WaveletCoeffNd, wavelet: Union[Wavelet, str], *, axes: _fwt.AxisHint = None) -> torch.Tensor:
    """Inverse of :func:`wavedec3` (src/ptwt/conv_transform_3.py:148-251)."""
    if len(coeffs) == 0 or not isinstance(coeffs[0], torch.Tensor):
        raise ValueError("First element of coeffs must be the approximation coefficient tensor.")
    levels = _fwt.unpack_dict_levels(coeffs, 3, "wavedec3")
    return _fwt.synthesis(coeffs[0], levels, wavelet, axes, 3, separable=False)
