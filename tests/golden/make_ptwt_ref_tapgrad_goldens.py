"""Golden gradients w.r.t. the FILTER TAPS from the reference itself (ptwt at /root/reference, imported with the
PyWavelets stand-in of tests/golden/_stubs): the four taps enter as leaf tensors (a 4-tuple is an accepted wavelet form,
src/ptwt/_util.py:115-121) and ATen autograd differentiates the reference's conv path w.r.t. them.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_ptwt_ref_tapgrad_goldens.py

loss_a = sum_i <w_i, c_i> (analysis), loss_s = <w_y, waverec(c)> (synthesis), w(t) = cos(0.37 arange + i)."""
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


def weight(t, i):
    return torch.cos(0.37 * torch.arange(t.numel(), dtype=torch.float64) + i).reshape(t.shape)


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


def case(fn, rec, shape, wavelet, seed, **kw):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(*shape, generator=g, dtype=torch.float64)
    bank = pywt.Wavelet(wavelet).filter_bank
    taps = [torch.tensor(list(t), dtype=torch.float64, requires_grad=True) for t in bank]
    wt = WaveletTensorTuple(*taps)
    coeffs = getattr(ptwt, fn)(x, wt, **kw)
    fl = flat(coeffs)
    loss = sum((weight(t, i) * t).sum() for i, t in enumerate(fl))
    g_dec = torch.autograd.grad(loss, taps[:2], retain_graph=True)
    rkw = {k: v for k, v in kw.items() if k in ("axis", "axes")}
    y = getattr(ptwt, rec)(coeffs, wt, **rkw)
    g_all = torch.autograd.grad((weight(y, 7) * y).sum(), taps, allow_unused=True)
    key = "t%03d" % len(index)
    store[key + "_x"] = x.numpy()
    store[key + "_gdec_lo"], store[key + "_gdec_hi"] = g_dec[0].numpy(), g_dec[1].numpy()
    for name, t in zip(("dec_lo", "dec_hi", "rec_lo", "rec_hi"), g_all):
        store["%s_gall_%s" % (key, name)] = t.numpy()
    kwj = {k: (list(v) if isinstance(v, tuple) else v) for k, v in kw.items()}
    index.append(dict(key=key, fn=fn, rec=rec, shape=list(shape), wavelet=wavelet, kw=kwj))


seed = 0
for mode in ("reflect", "zero", "constant", "periodic", "symmetric"):
    for wavelet in ("haar", "db3"):
        seed += 1
        case("wavedec", "waverec", (2, 37), wavelet, seed, mode=mode, level=2)
        case("wavedec2", "waverec2", (2, 21, 26), wavelet, seed, mode=mode, level=2)
        case("wavedec3", "waverec3", (1, 13, 14, 15), wavelet, seed, mode=mode, level=1)
    case("fswavedec2", "fswaverec2", (2, 22, 19), "db2", seed, mode=mode, level=2)
case("fswavedec3", "fswaverec3", (1, 12, 11, 13), "db2", 77, mode="reflect", level=1)
case("wavedec2", "waverec2", (2, 40, 44), "bior2.2", 78, mode="symmetric", level=2)
case("wavedec", "waverec", (3, 5, 64), "sym4", 79, mode="reflect", level=3, axis=-1)

# stationary transform: same loss construction (a list of tensors in, one tensor out)
for k, (shape, wavelet, level) in enumerate((((2, 64), "db2", 3), ((3, 48), "haar", 2), ((1, 96), "sym4", None))):
    case("swt", "iswt", shape, wavelet, 90 + k, level=level)


def packet_case(dim, shape, wavelet, mode, maxlevel, seed, **kw):
    """Packet trees: loss_a over the leaves of maxlevel (natural order), loss_s over reconstruct()'s root."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(*shape, generator=g, dtype=torch.float64)
    taps = [torch.tensor(list(t), dtype=torch.float64, requires_grad=True) for t in pywt.Wavelet(wavelet).filter_bank]
    wt = WaveletTensorTuple(*taps)
    cls = ptwt.WaveletPacket if dim == 1 else ptwt.WaveletPacket2D
    wp = cls(x, wt, mode=mode, maxlevel=maxlevel, **kw)
    keys = wp.get_level(maxlevel, "natural")
    loss = sum((weight(wp[k], i) * wp[k]).sum() for i, k in enumerate(keys))
    g_dec = torch.autograd.grad(loss, taps[:2], retain_graph=True)
    wp.reconstruct()
    y = wp[""]
    g_all = torch.autograd.grad((weight(y, 7) * y).sum(), taps, allow_unused=True)
    key = "t%03d" % len(index)
    store[key + "_x"] = x.numpy()
    store[key + "_gdec_lo"], store[key + "_gdec_hi"] = g_dec[0].numpy(), g_dec[1].numpy()
    for name, t in zip(("dec_lo", "dec_hi", "rec_lo", "rec_hi"), g_all):
        store["%s_gall_%s" % (key, name)] = t.numpy()
    kwj = {k: (list(v) if isinstance(v, tuple) else v) for k, v in kw.items()}
    index.append(dict(key=key, fn="packet%d" % dim, rec="reconstruct", shape=list(shape), wavelet=wavelet,
                      kw=dict(mode=mode, maxlevel=maxlevel, **kwj), keys=keys))


packet_case(1, (2, 67), "db3", "reflect", 3, 95)
packet_case(1, (3, 64), "haar", "periodic", 3, 96)
packet_case(2, (2, 35, 38), "db2", "symmetric", 2, 97)
packet_case(2, (1, 32, 32), "haar", "zero", 2, 98)
packet_case(2, (2, 35, 38), "db2", "constant", 2, 99, separable=True)

out = os.path.join(HERE, "ptwt_ref_tapgrads.npz")
np.savez_compressed(out, index=json.dumps(index), **store)
print("wrote", out, len(index), "cases", os.path.getsize(out) // 1024, "KiB")
