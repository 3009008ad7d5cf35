"""Golden SECOND-ORDER gradients with a learnable filter bank from the reference itself (ptwt at /root/reference, imported with the
PyWavelets stand-in of tests/golden/_stubs): the four taps are leaf tensors (src/ptwt/_util.py:115-121) and ATen's autograd
differentiates the reference's conv path twice (create_graph=True) — the mixed terms data x taps, taps x taps, upstream-gradient x taps.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_ptwt_ref_tapgrad2_goldens.py

Analysis:  f = sum_i <w_i, c_i^2> / 2;  g_x = df/dx and t = df/d(dec_lo, dec_hi), both with a graph;
           s1 = <g_x, v> + <t_lo, u_lo> + <t_hi, u_hi>  ->  ds1/dx, ds1/d dec_lo, ds1/d dec_hi.
Synthesis: f = <w_y, y^2> / 2, y = rec(leaves);  g_c = df/d leaves, t = df/d(rec_lo, rec_hi) with a graph;
           s2 = sum_i <g_c_i, v_i> + <t_lo, u_lo> + <t_hi, u_hi>  ->  ds2/d leaves, ds2/d rec_lo, ds2/d rec_hi.
w(t, i) = cos(0.37 arange + i); v, u = cos(0.53 arange + k)."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, "_stubs"))
sys.path.insert(0, "/root/reference/src")

import numpy as np  # noqa: E402
import pywt  # noqa: E402  (the stand-in)
import torch  # noqa: E402

import ptwt  # noqa: E402
from ptwt.constants import WaveletTensorTuple  # noqa: E402

store, index = {}, []


def weight(t, i, f=0.37):
    return torch.cos(f * torch.arange(t.numel(), dtype=torch.float64) + i).reshape(t.shape)


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
    bank = pywt.Wavelet(wavelet).filter_bank
    taps = [torch.tensor(list(t), dtype=torch.float64, requires_grad=True) for t in bank]
    wt = WaveletTensorTuple(*taps)
    key = "s%03d" % len(index)
    # ---- analysis
    fl = flat(getattr(ptwt, fn)(x, wt, **kw))
    f = sum((weight(t, i) * t.square()).sum() for i, t in enumerate(fl)) / 2
    g_x, t_lo, t_hi = torch.autograd.grad(f, [x, taps[0], taps[1]], create_graph=True)
    s1 = (g_x * weight(g_x, 1, 0.53)).sum() + (t_lo * weight(t_lo, 2, 0.53)).sum() + (t_hi * weight(t_hi, 3, 0.53)).sum()
    d = torch.autograd.grad(s1, [x, taps[0], taps[1]])
    store[key + "_x"] = x.detach().numpy()
    store[key + "_a_dx"], store[key + "_a_dlo"], store[key + "_a_dhi"] = (t.numpy() for t in d)
    # ---- synthesis
    coeffs = getattr(ptwt, fn)(x.detach(), pywt.Wavelet(wavelet), **kw)
    leaves = [t.detach().clone().requires_grad_(True) for t in flat(coeffs)]
    rkw = {k: v for k, v in kw.items() if k in ("axis", "axes")}
    y = getattr(ptwt, rec)(rebuild(coeffs, leaves), wt, **rkw)
    f = (weight(y, 7) * y.square()).sum() / 2
    grads = torch.autograd.grad(f, leaves + [taps[2], taps[3]], create_graph=True)
    s2 = sum((gc * weight(gc, 4 + i, 0.53)).sum() for i, gc in enumerate(grads[:-2]))
    s2 = s2 + (grads[-2] * weight(grads[-2], 2, 0.53)).sum() + (grads[-1] * weight(grads[-1], 3, 0.53)).sum()
    d2 = torch.autograd.grad(s2, leaves + [taps[2], taps[3]])
    for i, t in enumerate(d2[:-2]):
        store["%s_s_dc%d" % (key, i)] = t.numpy()
    store[key + "_s_dlo"], store[key + "_s_dhi"] = d2[-2].numpy(), d2[-1].numpy()
    kwj = {k: (list(v) if isinstance(v, tuple) else v) for k, v in kw.items()}
    index.append(dict(key=key, fn=fn, rec=rec, shape=list(shape), wavelet=wavelet, kw=kwj, ncoef=len(leaves)))


seed = 0
for mode in ("reflect", "zero", "periodic", "symmetric", "constant"):
    seed += 1
    case("wavedec", "waverec", (2, 37), "db3", seed, mode=mode, level=2)
    case("wavedec2", "waverec2", (2, 21, 26), "db2", seed, mode=mode, level=2)
case("wavedec3", "waverec3", (1, 13, 14, 15), "db2", 11, mode="reflect", level=1)
case("wavedec3", "waverec3", (1, 12, 14, 10), "haar", 12, mode="zero", level=2)
case("fswavedec2", "fswaverec2", (2, 22, 19), "db2", 13, mode="symmetric", level=2)
case("fswavedec3", "fswaverec3", (1, 12, 11, 13), "db2", 14, mode="reflect", level=1)
case("wavedec2", "waverec2", (2, 40, 44), "bior2.2", 15, mode="symmetric", level=2)
case("wavedec", "waverec", (3, 5, 64), "sym4", 16, mode="reflect", level=3, axis=-1)
# stationary transform (a list of tensors in, one tensor out)
case("swt", "iswt", (2, 64), "db2", 17, level=3)
case("swt", "iswt", (3, 48), "haar", 18, level=2)
case("swt", "iswt", (1, 96), "sym4", 19, level=None)

out = os.path.join(HERE, "ptwt_ref_tapgrads2.npz")
np.savez_compressed(out, index=json.dumps(index), **store)
print("wrote", out, len(index), "cases", os.path.getsize(out) // 1024, "KiB")
