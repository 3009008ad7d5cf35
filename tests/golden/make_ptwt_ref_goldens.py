"""Generate golden vectors from the REFERENCE ITSELF (ptwt at /root/reference), imported in the build
container with the PyWavelets stand-in of tests/golden/_stubs (the interpreter that has torch lacks pywt).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_ptwt_ref_goldens.py

/root/reference does not exist on the GPU box, so its outputs travel as this fixture
(tests/golden/ptwt_ref.npz).  Covers what the pywt goldens cannot: the separable API, synthesis outputs
(incl. the odd-length "+1 sample" behaviour), non-default axes, folded batch dims, float32 arithmetic of the
reference's dense conv path, and BASELINE config 1 (Haar, N=4096, fp64, 12 levels).
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, "_stubs"))
sys.path.insert(0, "/root/reference/src")

import numpy as np  # noqa: E402
import torch  # noqa: E402

import ptwt  # noqa: E402

torch.set_num_threads(4)
store, index = {}, []


def put(key, name, t):
    store["%s_%s" % (key, name)] = t.detach().cpu().numpy()


def flat(coeffs):
    """Flatten any ptwt coefficient container into [(name, tensor)]."""
    out = [("a", coeffs[0])]
    for i, c in enumerate(coeffs[1:]):
        if isinstance(c, torch.Tensor):
            out.append(("%d" % i, c))
        elif isinstance(c, dict):
            out.extend(("%d_%s" % (i, k), v) for k, v in c.items())
        else:
            out.extend(("%d_%s" % (i, n), v) for n, v in zip("hvd", c))
    return out


def case(fn, rec, shape, wavelet, dtype, seed, **kw):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(*shape, generator=g, dtype=torch.float64).to(dtype)
    coeffs = getattr(ptwt, fn)(x, wavelet, **kw)
    rkw = {}
    if "axis" in kw:
        rkw["axis"] = kw["axis"]
    if "axes" in kw:
        rkw["axes"] = kw["axes"]
    # flatten BEFORE synthesis: the separable reference inserts the approximation into the caller's
    # level dicts as a side effect (src/ptwt/separable_conv_transform.py:181-182)
    flat_coeffs = flat(coeffs)
    y = getattr(ptwt, rec)(coeffs, wavelet, **rkw)
    key = "r%03d" % len(index)
    names = []
    put(key, "x", x)
    put(key, "rec", y)
    for n, t in flat_coeffs:
        put(key, n, t)
        names.append(n)
    kwj = {k: (list(v) if isinstance(v, tuple) else v) for k, v in kw.items()}
    index.append(dict(key=key, fn=fn, rec=rec, shape=list(shape), wavelet=wavelet,
                      dtype=str(dtype).split(".")[-1], kw=kwj, names=names))


f64, f32 = torch.float64, torch.float32
seed = 0
# BASELINE config 1: Haar, N=4096, batch 1, fp64, all 12 levels
case("wavedec", "waverec", (1, 4096), "haar", f64, 100)
for mode in ("reflect", "zero", "constant", "periodic", "symmetric"):
    for dt in (f64, f32):
        seed += 1
        case("wavedec", "waverec", (3, 65), "db4", dt, seed, mode=mode, level=2)
        case("wavedec2", "waverec2", (2, 33, 30), "db4", dt, seed, mode=mode, level=2)
        if dt == f32:
            continue
        case("wavedec", "waverec", (2, 3, 50), "sym5", dt, seed, mode=mode)
        case("wavedec", "waverec", (40, 3), "db2", dt, seed, mode=mode, level=2, axis=0)
        case("wavedec2", "waverec2", (21, 2, 26), "db2", dt, seed, mode=mode, level=2, axes=(0, 2))
        case("wavedec2", "waverec2", (2, 2, 16, 19), "bior2.2", dt, seed, mode=mode, level=1)
        case("wavedec3", "waverec3", (1, 9, 8, 11), "db2", dt, seed, mode=mode, level=2)
        case("wavedec3", "waverec3", (6, 7, 2, 9), "haar", dt, seed, mode=mode, level=1, axes=(0, 1, 3))
        case("fswavedec2", "fswaverec2", (2, 29, 32), "db3", dt, seed, mode=mode, level=2)
        case("fswavedec2", "fswaverec2", (18, 2, 17), "db2", dt, seed, mode=mode, level=1, axes=(2, 0))
        case("fswavedec3", "fswaverec3", (1, 8, 9, 10), "db2", dt, seed, mode=mode, level=2)
# default-mode / default-level calls
case("wavedec2", "waverec2", (32, 32), "db2", f64, 200)
case("wavedec3", "waverec3", (16, 16, 16), "haar", f64, 201)
case("fswavedec2", "fswaverec2", (1, 32, 32), "db2", f64, 202)
case("fswavedec3", "fswaverec3", (1, 16, 16, 16), "haar", f64, 203)
# long filter (sym16, L=32): symmetric mode tolerates pad > N
case("wavedec2", "waverec2", (1, 40, 44), "sym16", f64, 204, mode="symmetric", level=1)
case("fswavedec2", "fswaverec2", (1, 40, 36), "sym16", f64, 205, mode="reflect", level=1)

store["index"] = np.array(json.dumps(index))
np.savez_compressed(os.path.join(HERE, "ptwt_ref.npz"), **store)
print("reference cases:", len(index))
