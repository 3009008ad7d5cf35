"""ptwt_amd — MI355X-native padded-convolution fast wavelet transforms behind the ptwt API.

Drop-in for the convolution-FWT path of v0lta/PyTorch-Wavelet-Toolbox (``ptwt``): the ten functions below
keep ptwt's signatures, defaults, return containers and error behaviour (reference
src/ptwt/__init__.py:12-19); every decomposition / reconstruction level runs as a hand-written HIP kernel
for gfx950 reached through the C ABI of ``libmifwt.so`` (include/mifwt.h).  Tensors must live on a ROCm
device; there is no CPU fallback.

    import ptwt_amd as ptwt
    coeffs = ptwt.wavedec2(x.cuda(), "db4", level=3)
"""
from .constants import (
    Wavelet,
    WaveletCoeff2d,
    WaveletCoeff2dSeparable,
    WaveletCoeffNd,
    WaveletDetailDict,
    WaveletDetailTuple2d,
    WaveletTensorTuple,
    set_half_storage,
    half_storage,
)
from .conv_transform import wavedec, waverec
from .conv_transform_2 import wavedec2, waverec2
from .conv_transform_3 import wavedec3, waverec3
from .packets import WaveletPacket, WaveletPacket2D
from .stationary_transform import iswt, swt
from .separable_conv_transform import fswavedec2, fswavedec3, fswaverec2, fswaverec3
from .graphs import CapturedCall, capture
from ._wavelets import set_device_taps

__version__ = "0.1.0"

__all__ = [
    "Wavelet",
    "WaveletDetailTuple2d",
    "WaveletCoeff2d",
    "WaveletCoeff2dSeparable",
    "WaveletCoeffNd",
    "WaveletDetailDict",
    "WaveletTensorTuple",
    "wavedec",
    "waverec",
    "wavedec2",
    "waverec2",
    "wavedec3",
    "waverec3",
    "fswavedec2",
    "fswavedec3",
    "fswaverec2",
    "fswaverec3",
    "set_half_storage",
    "half_storage",
    "set_device_taps",
    "WaveletPacket",
    "swt",
    "iswt",
    "WaveletPacket2D",
    "capture",
    "CapturedCall",
]
