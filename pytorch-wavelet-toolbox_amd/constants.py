"""Public types of the conv-FWT path, name-compatible with ``ptwt.constants`` (reference
src/ptwt/constants.py:27-253) so annotations and ``isinstance`` checks written against ptwt keep working.
"""
from __future__ import annotations

import contextlib
import contextvars
from typing import Dict, Literal, NamedTuple, Protocol, Sequence, Tuple, Union

import torch

__all__ = [
    "BoundaryMode",
    "SUPPORTED_DTYPES",
    "set_half_storage",
    "half_storage",
    "Wavelet",
    "WaveletCoeff1d",
    "WaveletCoeff2d",
    "WaveletCoeff2dSeparable",
    "WaveletCoeffNd",
    "WaveletDetailDict",
    "WaveletDetailTuple2d",
    "WaveletTensorTuple",
]

#: dtypes the reference accepts (src/ptwt/constants.py:27); anything else raises ``ValueError``.
SUPPORTED_DTYPES = {torch.float32, torch.float64}

#: Engine extension (NOT in the reference, which raises ``ValueError`` for it): float16 STORAGE with float32
#: arithmetic (C ABI ``MIFWT_F16``).  Off by default so that error behaviour matches the reference.  Switched on for a
#: scope with :func:`half_storage` (a context manager over a ``contextvars`` variable: local to the thread / task, so
#: one caller's fp16 workload does not change what another thread's call accepts); :func:`set_half_storage` sets the
#: process-wide DEFAULT the scoped value falls back to (BASELINE.json configs[4] is an fp16 workload).
_half_storage_default = False
_half_storage_scoped: "contextvars.ContextVar[object]" = contextvars.ContextVar("ptwt_amd_half_storage", default=None)


def set_half_storage(enabled: bool) -> None:
    """Process-wide default: accept ``torch.float16`` inputs (coefficients are returned in float16, accumulated in
    float32).  Prefer ``with ptwt_amd.half_storage(): ...`` — scoped, thread-local."""
    global _half_storage_default
    _half_storage_default = bool(enabled)


@contextlib.contextmanager
def half_storage(enabled: bool = True):
    """``with half_storage():`` — calls in this scope (this thread / task only) accept ``torch.float16`` tensors."""
    token = _half_storage_scoped.set(bool(enabled))
    try:
        yield
    finally:
        _half_storage_scoped.reset(token)


def supported_dtypes():
    scoped = _half_storage_scoped.get()
    on = _half_storage_default if scoped is None else scoped
    return SUPPORTED_DTYPES | {torch.float16} if on else SUPPORTED_DTYPES

#: boundary rules (src/ptwt/constants.py:85): zero | constant (edge replicate) | reflect (whole-sample
#: mirror) | periodic | symmetric (half-sample mirror)
BoundaryMode = Literal["constant", "zero", "reflect", "periodic", "symmetric"]


class Wavelet(Protocol):
    """Anything shaped like ``pywt.Wavelet``: four tap sequences and a length."""

    name: str
    dec_lo: Sequence[float]
    dec_hi: Sequence[float]
    rec_lo: Sequence[float]
    rec_hi: Sequence[float]
    dec_len: int
    rec_len: int
    filter_bank: Tuple[Sequence[float], Sequence[float], Sequence[float], Sequence[float]]

    def __len__(self) -> int:
        return len(self.dec_lo)


class WaveletTensorTuple(NamedTuple):
    """Filter bank as four tensors (the form the reference uses under ``torch.jit.trace``)."""

    dec_lo: torch.Tensor
    dec_hi: torch.Tensor
    rec_lo: torch.Tensor
    rec_hi: torch.Tensor

    @property
    def dec_len(self) -> int:
        return len(self.dec_lo)

    @property
    def rec_len(self) -> int:
        return len(self.rec_lo)

    @property
    def filter_bank(self) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        return self

    @classmethod
    def from_wavelet(cls, wavelet: Wavelet, dtype: torch.dtype) -> "WaveletTensorTuple":
        return cls(*(torch.tensor(list(t), dtype=dtype) for t in (wavelet.dec_lo, wavelet.dec_hi, wavelet.rec_lo, wavelet.rec_hi)))


class WaveletDetailTuple2d(NamedTuple):
    """``(H, V, D)`` detail bands of one 2-D level: H = high-pass along the first transformed axis
    (pywt ``'da'``), V = ``'ad'``, D = ``'dd'`` (src/ptwt/_util.py:901-905)."""

    horizontal: torch.Tensor
    vertical: torch.Tensor
    diagonal: torch.Tensor


#: ``[cA_n, cD_n, ..., cD_1]``
WaveletCoeff1d = Sequence[torch.Tensor]
#: ``{"aad": ..., ..., "ddd": ...}`` — key char i <-> transformed axis i, 'a' low-pass, 'd' high-pass
WaveletDetailDict = Dict[str, torch.Tensor]
#: ``(cA_n, (H,V,D)_n, ..., (H,V,D)_1)``
WaveletCoeff2d = Tuple[Union[torch.Tensor, WaveletDetailTuple2d], ...]
#: ``(cA_n, {..}_n, ..., {..}_1)``
WaveletCoeffNd = Tuple[Union[torch.Tensor, WaveletDetailDict], ...]
WaveletCoeff2dSeparable = WaveletCoeffNd
