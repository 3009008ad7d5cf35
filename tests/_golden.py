"""Helpers to read the committed golden fixtures (tests/golden/*.npz)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_CACHE = {}


def load(name):
    if name not in _CACHE:
        z = np.load(os.path.join(GOLDEN, name))
        _CACHE[name] = (z, json.loads(str(z["index"])))
    return _CACHE[name]


def relerr(got, want):
    """Norm-wise relative error (the metric SURVEY.md §8c prescribes; element-wise is meaningless near 0)."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    den = np.linalg.norm(want.ravel())
    num = np.linalg.norm((got - want).ravel())
    return num / den if den > 0 else num


def pywt1d_cases():
    z, idx = load("pywt_wavedec1d.npz")
    return idx


def pywt2d_cases():
    z, idx = load("pywt_wavedec2d.npz")
    return idx


def pywt3d_cases():
    z, idx = load("pywt_wavedec3d.npz")
    return idx


def ref_cases():
    z, idx = load("ptwt_ref.npz")
    return idx


def flatten_coeffs(coeffs):
    """[(name, array)] in the naming of make_ptwt_ref_goldens.flat."""
    out = [("a", coeffs[0])]
    for i, c in enumerate(coeffs[1:]):
        if isinstance(c, dict):
            out.extend(("%d_%s" % (i, k), v) for k, v in c.items())
        elif isinstance(c, (tuple, list)):
            out.extend(("%d_%s" % (i, n), v) for n, v in zip("hvd", c))
        else:
            out.append(("%d" % i, c))
    return out
