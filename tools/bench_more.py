#!/usr/bin/env python
"""Secondary timings (not the headline): synthesis and the non-2-D paths, whole call, HIP-event timed.
Prints one JSON line per case: ms per call, Msamples/s, and algorithmic (compulsory) GB/s."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import __graft_entry__ as entry  # noqa: E402

entry.build(verbose=False)
import ptwt_amd  # noqa: E402

dev = torch.device("cuda:0")
CASES = [
    # name, analysis fn, synthesis fn, shape, wavelet, level, mode, dtype
    ("config2 2-D db4 L3", "wavedec2", "waverec2", (64, 1024, 1024), "db4", 3, "reflect", torch.float32),
    ("config3 3-D db2 L3", "wavedec3", "waverec3", (8, 256, 256, 256), "db2", 3, "zero", torch.float32),
    ("1-D db5 L10 (ref speed test)", "wavedec", "waverec", (32, 1000000), "db5", 10, "periodic", torch.float32),
    ("2-D db4 L3 f64", "wavedec2", "waverec2", (32, 1024, 1024), "db4", 3, "reflect", torch.float64),
    ("2-D sym16 L3 f32 (L=32)", "fswavedec2", "fswaverec2", (16, 2048, 2048), "sym16", 3, "reflect", torch.float32),
]
if len(sys.argv) > 1:
    CASES = [c for c in CASES if any(k in c[0] for k in sys.argv[1:])]


def timeit(fn, n=10):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


for name, afn, sfn, shape, wavelet, level, mode, dtype in CASES:
    xs = [torch.randn(*shape, device=dev, dtype=dtype) for _ in range(3)]
    esz = xs[0].element_size()
    i = [0]

    def fwd():
        i[0] += 1
        return getattr(ptwt_amd, afn)(xs[i[0] % 3], wavelet, mode=mode, level=level)

    cs = [getattr(ptwt_amd, afn)(x, wavelet, mode=mode, level=level) for x in xs]

    def inv():
        i[0] += 1
        return getattr(ptwt_amd, sfn)(cs[i[0] % 3], wavelet)

    def numel(c):
        if isinstance(c, torch.Tensor):
            return c.numel()
        if isinstance(c, dict):
            return sum(v.numel() for v in c.values())
        return sum(numel(v) for v in c)

    comp = (xs[0].numel() + numel(cs[0])) * esz
    for tag, f in (("analysis", fwd), ("synthesis", inv)):
        ms = timeit(f)
        print(json.dumps({"case": name, "dir": tag, "fn": afn if tag == "analysis" else sfn, "shape": shape, "dtype": str(dtype),
                          "ms": round(ms, 4), "Msamples/s": round(xs[0].numel() / ms / 1e3, 1),
                          "GBps_compulsory": round(comp / ms / 1e6, 1), "frac_8TBps": round(comp / ms / 1e6 / 8000, 4)}), flush=True)
    del xs, cs
    torch.cuda.empty_cache()
