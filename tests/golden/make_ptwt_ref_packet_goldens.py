"""Golden wavelet-packet trees from the REFERENCE (ptwt.WaveletPacket / WaveletPacket2D at /root/reference, imported
with the PyWavelets stand-in of tests/golden/_stubs).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_ptwt_ref_packet_goldens.py

Per case: the input, every node of the deepest level (natural order), and the reconstruction after the leaves
were scaled by 0.5 (exercises reconstruct() incl. the odd-length crop)."""
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


def case(dim, shape, wavelet, mode, maxlevel, seed, **kw):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(*shape, generator=g, dtype=torch.float64)
    cls = ptwt.WaveletPacket if dim == 1 else ptwt.WaveletPacket2D
    wp = cls(x, wavelet, mode=mode, maxlevel=maxlevel, **kw)
    keys = wp.get_level(maxlevel, "natural")
    key = "p%03d" % len(index)
    store[key + "_x"] = x.numpy()
    for k in keys:
        store["%s_n_%s" % (key, k)] = wp[k].numpy()
    for k in keys:
        wp[k] = 0.5 * wp[k]
    wp.reconstruct()
    store[key + "_rec"] = wp[""].numpy()
    kwj = {k: (list(v) if isinstance(v, tuple) else v) for k, v in kw.items()}
    index.append(dict(key=key, dim=dim, shape=list(shape), wavelet=wavelet, mode=mode, maxlevel=maxlevel, kw=kwj, keys=keys))


seed = 0
for mode in ("reflect", "zero", "constant", "periodic", "symmetric"):
    seed += 1
    case(1, (2, 67), "db3", mode, 3, seed)
    case(1, (3, 64), "haar", mode, 4, seed)
    case(2, (2, 35, 38), "db2", mode, 2, seed)
    case(2, (1, 32, 32), "haar", mode, 3, seed)
    case(2, (2, 35, 38), "db2", mode, 2, seed, separable=True)
case(1, (2, 40, 3), "sym4", "reflect", 2, 50, axis=1)
case(2, (2, 30, 3, 34), "db2", "symmetric", 2, 51, axes=(1, 3))
case(2, (31, 33), "db3", "reflect", 2, 52)
case(1, (50,), "db2", "zero", 3, 53)

out = os.path.join(HERE, "ptwt_ref_packets.npz")
np.savez_compressed(out, index=json.dumps(index), **store)
print("wrote", out, len(index), "cases", os.path.getsize(out) // 1024, "KiB")
