"""Golden GRADIENTS from the reference itself (ptwt at /root/reference, imported with the PyWavelets stand-in of
tests/golden/_stubs): autograd through the reference's F.pad + F.conv*d / conv_transpose*d path.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_ptwt_ref_grad_goldens.py

For every case: x (fp64), the gradient of  loss_a = sum_i <w_i, c_i>  w.r.t. x  (analysis backward) and the
gradients of  loss_s = <w_y, waverec(c)>  w.r.t. every coefficient tensor (synthesis backward), with the fixed
weights  w(t) = cos(0.37 * arange(t.numel()) + i).reshape(t.shape)  (i = position in the flattened container), so
that the test can rebuild them without storing them.
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
    return torch.cos(0.37 * torch.arange(t.numel(), dtype=torch.float64) + i).reshape(t.shape).to(t.dtype)


def flat(coeffs):
    out = [coeffs[0]]
    for c in coeffs[1:]:
        if isinstance(c, torch.Tensor):
            out.append(c)
        elif isinstance(c, dict):
            out.extend(c.values())
        else:
            out.extend(c)
    return out


def rebuild(coeffs, leaves):
    """Same container as ``coeffs`` with the tensors replaced by ``leaves`` (flattening order)."""
    it = iter(leaves)
    out = [next(it)]
    for c in coeffs[1:]:
        if isinstance(c, torch.Tensor):
            out.append(next(it))
        elif isinstance(c, dict):
            out.append({k: next(it) for k in c})
        else:
            out.append(type(c)(*[next(it) for _ in c]))
    return out if isinstance(coeffs, list) else tuple(out)


def case(fn, rec, shape, wavelet, seed, **kw):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(*shape, generator=g, dtype=torch.float64, requires_grad=True)
    coeffs = getattr(ptwt, fn)(x, wavelet, **kw)
    fl = flat(coeffs)
    loss = sum((weight(t, i) * t).sum() for i, t in enumerate(fl))
    (gx,) = torch.autograd.grad(loss, x)
    leaves = [t.detach().clone().requires_grad_(True) for t in fl]
    rkw = {k: v for k, v in kw.items() if k in ("axis", "axes")}
    y = getattr(ptwt, rec)(rebuild(coeffs, leaves), wavelet, **rkw)
    gl = torch.autograd.grad((weight(y, 7) * y).sum(), leaves)
    key = "g%03d" % len(index)
    store[key + "_x"] = x.detach().numpy()
    store[key + "_gx"] = gx.numpy()
    for i, t in enumerate(gl):
        store["%s_gc%d" % (key, i)] = t.numpy()
    kwj = {k: (list(v) if isinstance(v, tuple) else v) for k, v in kw.items()}
    index.append(dict(key=key, fn=fn, rec=rec, shape=list(shape), wavelet=wavelet, kw=kwj, ncoef=len(fl)))


seed = 0
for mode in ("reflect", "zero", "constant", "periodic", "symmetric"):
    for wavelet in ("haar", "db3", "sym4"):
        seed += 1
        case("wavedec", "waverec", (2, 37), wavelet, seed, mode=mode, level=2)
        case("wavedec2", "waverec2", (2, 21, 26), wavelet, seed, mode=mode, level=2)
        case("wavedec3", "waverec3", (1, 13, 14, 15), wavelet, seed, mode=mode, level=1)
    case("fswavedec2", "fswaverec2", (2, 22, 19), "db2", seed, mode=mode, level=2)
    case("fswavedec3", "fswaverec3", (1, 12, 11, 13), "db2", seed, mode=mode, level=1)
case("wavedec2", "waverec2", (3, 18, 2, 20), "db2", 99, mode="reflect", level=1, axes=(1, 3))
case("wavedec", "waverec", (4, 30, 3), "db2", 98, mode="symmetric", level=2, axis=1)
case("wavedec", "waverec", (1, 4), "db4", 97, mode="symmetric", level=1)  # pad wraps more than once

out = os.path.join(HERE, "ptwt_ref_grads.npz")
np.savez_compressed(out, index=json.dumps(index), **store)
print("wrote", out, len(index), "cases", os.path.getsize(out) // 1024, "KiB")
