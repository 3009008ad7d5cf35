"""Placeholder so `from pywt._functions import scale2frequency` in unrelated reference modules imports."""


def scale2frequency(wavelet, scale, precision=8):
    raise NotImplementedError("stub")


def integrate_wavelet(wavelet, precision=8):
    raise NotImplementedError("stub")
