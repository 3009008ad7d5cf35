"""Golden stationary-transform vectors from the REFERENCE (ptwt.swt / ptwt.iswt at /root/reference, imported with the
PyWavelets stand-in of tests/golden/_stubs), incl. gradients of the reference's autograd.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_ptwt_ref_swt_goldens.py
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

store, index = {}, []


def weight(t, i):
    return torch.cos(0.37 * torch.arange(t.numel(), dtype=torch.float64) + i).reshape(t.shape)


def case(shape, wavelet, level, seed, **kw):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(*shape, generator=g, dtype=torch.float64, requires_grad=True)
    c = ptwt.swt(x, wavelet, level, **kw)
    key = "s%03d" % len(index)
    store[key + "_x"] = x.detach().numpy()
    for i, t in enumerate(c):
        store["%s_c%d" % (key, i)] = t.detach().numpy()
    (gx,) = torch.autograd.grad(sum((weight(t, i) * t).sum() for i, t in enumerate(c)), x)
    store[key + "_gx"] = gx.numpy()
    leaves = [t.detach().clone().requires_grad_(True) for t in c]
    y = ptwt.iswt(leaves, wavelet, **kw)
    store[key + "_rec"] = y.detach().numpy()
    gl = torch.autograd.grad((weight(y, 7) * y).sum(), leaves)
    for i, t in enumerate(gl):
        store["%s_gc%d" % (key, i)] = t.numpy()
    index.append(dict(key=key, shape=list(shape), wavelet=wavelet, level=level, kw=kw, ncoef=len(c)))


seed = 0
for wavelet in ("haar", "db2", "db4", "sym5", "db8", "bior2.2"):
    for shape, level in (((2, 64), 3), ((3, 96), None), ((1, 24), 2)):
        seed += 1
        case(shape, wavelet, level, seed)
case((2, 32, 3), "db3", 2, 90, axis=1)
case((2, 3, 48), "db2", 4, 91)          # dilation * L > N: the circular pad wraps more than once
case((40,), "db2", 3, 92)
# filters longer than the unrolled kernel instantiations (run-time tap loop): 22 .. 102 taps
for k, (wavelet, shape, level) in enumerate((("db11", (2, 64), 2), ("db12", (2, 128), 3), ("sym13", (1, 96), 2), ("coif4", (3, 64), 3),
                                             ("dmey", (2, 128), 2), ("db38", (1, 160), 2), ("coif17", (2, 256), 2))):
    case(shape, wavelet, level, 100 + k)

out = os.path.join(HERE, "ptwt_ref_swt.npz")
np.savez_compressed(out, index=json.dumps(index), **store)
print("wrote", out, len(index), "cases", os.path.getsize(out) // 1024, "KiB")
