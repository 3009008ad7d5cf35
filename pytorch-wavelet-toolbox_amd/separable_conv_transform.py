"""Separable N-D FWT: ``fswavedec2/3`` / ``fswaverec2/3`` (API of reference
src/ptwt/separable_conv_transform.py:187-446).

In the reference these run 3 (2-D) or 7 (3-D) single-axis ``wavedec`` calls per level with transposes in
between; numerically they equal ``wavedec2`` / ``wavedec3`` (the N-D filters are outer products of the 1-D
pair) and differ only in the return container and in how synthesis handles odd extents.  Here both API
families share the same fused per-level kernel.
"""
from __future__ import annotations

from typing import Optional, Union

import torch

from . import _fwt
from .constants import BoundaryMode, Wavelet, WaveletCoeff2dSeparable, WaveletCoeffNd

__all__ = ["fswavedec2", "fswavedec3", "fswaverec2", "fswaverec3"]


def _fswavedecn(data, wavelet, ndim, *, mode="reflect", level=None, axes=None) -> WaveletCoeffNd:
    layout, approx, bufs = _fwt.analysis(data, wavelet, mode, level, axes, ndim)
    return _fwt.pack_dict(layout, approx, bufs, _fwt._KEYS_FS[ndim])


def _fswaverecn(coeffs, wavelet, ndim, *, axes=None) -> torch.Tensor:
    if len(coeffs) == 0 or not isinstance(coeffs[0], torch.Tensor):
        raise ValueError("approximation tensor must be first in coefficient list.")
    if not all(isinstance(c, dict) for c in coeffs[1:]):
        raise ValueError("All entries after approximation tensor must be dicts.")
    levels = _fwt.unpack_dict_levels(coeffs, ndim, f"fswavedec{ndim}")
    return _fwt.synthesis(coeffs[0], levels, wavelet, axes, ndim, separable=True)


def fswavedec2(data: torch.Tensor, wavelet: Union[Wavelet, str], *, mode: BoundaryMode = "reflect",
               level: Optional[int] = None, axes: _fwt.AxisHint = None) -> WaveletCoeff2dSeparable:
    """``(cA_n, {"da","ad","dd"}_n, ...)`` — drop-in for ``ptwt.fswavedec2`` (:187-231)."""
    return _fswavedecn(data, wavelet, 2, mode=mode, level=level, axes=axes)


def fswavedec3(data: torch.Tensor, wavelet: Union[Wavelet, str], *, mode: BoundaryMode = "reflect",
               level: Optional[int] = None, axes: _fwt.AxisHint = None) -> WaveletCoeffNd:
    """``(cA_n, {"aad",...,"ddd"}_n, ...)`` — drop-in for ``ptwt.fswavedec3`` (:234-278)."""
    return _fswavedecn(data, wavelet, 3, mode=mode, level=level, axes=axes)


def fswaverec2(coeffs: WaveletCoeff2dSeparable, wavelet: Union[Wavelet, str], *, axes: _fwt.AxisHint = None) -> torch.Tensor:
    """Inverse of :func:`fswavedec2` (:281-313)."""
    return _fswaverecn(coeffs, wavelet, 2, axes=axes)


def fswaverec3(coeffs: WaveletCoeffNd, wavelet: Union[Wavelet, str], *, axes: _fwt.AxisHint = None) -> torch.Tensor:
    """Inverse of :func:`fswavedec3` (:316-348)."""
    return _fswaverecn(coeffs, wavelet, 3, axes=axes)
