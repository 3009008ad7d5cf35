#!/usr/bin/env python
"""Does the 256 MiB Infinity Cache absorb a re-used scratch buffer?  Times wavedec3 level 1 (fused planes + depth
pass, scratch = 4 planes per slice) on a shallow volume whose scratch (68 MB) fits, with inputs rotated through
> 256 MiB and every output kept alive (fresh memory per call), against the deep volume (scratch 545 MB)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as entry
entry.build(verbose=False)
import ptwt_amd
dev = torch.device("cuda:0")
for shape, nbuf in [((8, 256, 256, 256), 3), ((8, 32, 256, 256), 10), ((8, 16, 256, 256), 20), ((2, 32, 256, 256), 40)]:
    xs = [torch.randn(*shape, device=dev) for _ in range(nbuf)]
    keep = []
    for i in range(2):
        ptwt_amd.wavedec3(xs[i], "db2", level=1)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(nbuf):
        keep.append(ptwt_amd.wavedec3(xs[i], "db2", level=1))
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / nbuf
    n = xs[0].numel()
    print(json.dumps({"shape": shape, "ms": round(ms, 4), "ns_per_Msample": round(ms * 1e6 / (n / 1e6), 1),
                      "algorithmic_GBps": round(2 * 4 * n * 1.02 / ms / 1e6, 1)}))
    del xs, keep
    torch.cuda.empty_cache()
