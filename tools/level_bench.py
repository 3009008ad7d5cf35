#!/usr/bin/env python
"""Kernel-level A/B harness: times ONE analysis level (default: level 1 of BASELINE config 2) through the C
ABI with HIP events, interleaving variants in one process (rounds x variants), rotating input buffers.

    python tools/level_bench.py --rpc 8,16,32,64 --rounds 5
"""
import argparse
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import __graft_entry__ as entry  # noqa: E402

entry.build(verbose=False)
from ptwt_amd import _engine, _wavelets  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="64,1024,1024")
ap.add_argument("--wavelet", default="db4")
ap.add_argument("--mode", default="reflect")
ap.add_argument("--rpc", default="0", help="comma list of rows-per-chunk overrides (0 = library default)")
ap.add_argument("--depth", default="0", help="comma list of prefetch-depth overrides (0 = library default)")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--generic", action="store_true", help="also time the generic axis-pass path")
ap.add_argument("--inverse", action="store_true", help="time the synthesis level that reconstructs --shape instead")
ap.add_argument("--nt", default="0", help="comma list: 1 = nontemporal stores")
ap.add_argument("--tile", default="0", help="comma list: tile mode (1 = LDS-tile kernel, 2 = streaming kernel, 0 = auto)")
ap.add_argument("--tr", default="0", help="comma list: tile rows override (8 / 16)")
ap.add_argument("--pair", action="store_true", help="time the two-levels-per-launch kernel (levels 1+2) instead of one level")
args = ap.parse_args()

shape = tuple(int(v) for v in args.shape.split(","))
dev = torch.device("cuda:0")
taps = _wavelets.host_taps(args.wavelet)
flen = len(taps[0])
bufs = [torch.randn(*shape, device=dev) for _ in range(3)]
coef = [(n + flen - 1) // 2 for n in shape[1:]]
nb = 1 << (len(shape) - 1)
bytes_algo = 4 * shape[0] * (torch.Size(shape[1:]).numel() + nb * torch.Size(coef).numel())
eng = _engine.ENGINE
mode_id = _engine.MODE_IDS[args.mode]
if args.inverse:
    cbufs = [eng.analysis(b, taps[0], taps[1], mode_id) for b in bufs]
    out_ext = [2 * m - flen + 2 - (n % 2) for m, n in zip(coef, shape[1:])]

    def run(i):
        cb = cbufs[i % 3]
        return eng.synthesis(cb[:, 0], [cb[:, s] for s in range(1, nb)], taps[2], taps[3], out_ext)
elif args.pair:
    coef2 = [(n + flen - 1) // 2 for n in coef]
    bytes_algo = 4 * shape[0] * (torch.Size(shape[1:]).numel() + (nb - 1) * torch.Size(coef).numel() + nb * torch.Size(coef2).numel())

    def run(i):
        out = eng.analysis_pair(bufs[i % 3], taps[0], taps[1], mode_id)
        assert out is not None, "the library does not serve this geometry as a pair"
        return out
else:
    def run(i):
        return eng.analysis(bufs[i % 3], taps[0], taps[1], mode_id)

variants = [("tile=%s tr=%s rpc=%s depth=%s" % (c, n, r, d), int(r), 0, int(d), int(c), int(n)) for c in args.tile.split(",")
            for n in args.tr.split(",") for r in args.rpc.split(",") for d in args.depth.split(",")]
if args.generic:
    variants.append(("generic", 0, 1, 0, 0, 0))
results = {name: [] for name, _, _, _, _, _ in variants}
for rnd in range(args.rounds + 1):
    for name, rpc, gen, depth, coop, nt in variants:
        _engine.set_option(6, nt)
        _engine.set_option(1, rpc)
        _engine.set_option(0, gen)
        _engine.set_option(2, depth)
        _engine.set_option(5, coop)
        run(0)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(args.iters):
            run(i)
        e.record()
        torch.cuda.synchronize()
        if rnd > 0:
            results[name].append(s.elapsed_time(e) / args.iters)
_engine.set_option(1, 0)
_engine.set_option(0, 0)
_engine.set_option(2, 0)
_engine.set_option(5, 0)
_engine.set_option(6, 0)
for name, ts in results.items():
    med = statistics.median(ts)
    print(json.dumps({"variant": name, "shape": shape, "wavelet": args.wavelet, "ms_median": round(med, 4),
                      "ms_min": round(min(ts), 4), "GBps_algorithmic": round(bytes_algo / med / 1e6, 1),
                      "frac_8TBps": round(bytes_algo / med / 1e6 / 8000, 4)}))
