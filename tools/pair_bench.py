#!/usr/bin/env python
"""A/B harness for the two-levels-per-launch analysis kernel (mifwt_dwt2_fwd_pair): times the WHOLE wavedec2 call
(default BASELINE config 2) with HIP events, interleaving variants in one process, rotating input buffers.

    python tools/pair_bench.py --rows 0,4,6,8,12 --rounds 5
variant "single" = per-level kernels only (OPT_PAIR_MODE 2); "pair rows=R" = pair kernel, R level-2 rows per tile.
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
import ptwt_amd  # noqa: E402
from ptwt_amd import _engine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="64,1024,1024")
ap.add_argument("--wavelet", default="db4")
ap.add_argument("--mode", default="reflect")
ap.add_argument("--level", type=int, default=3)
ap.add_argument("--rows", default="8", help="tile-pair kernel: level-2 rows per tile")
ap.add_argument("--seg", default="32", help="rolling kernel: level-2 rows per segment")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--inverse", action="store_true", help="time waverec2 of the coefficients instead (single vs auto)")
ap.add_argument("--sync-stage", type=int, default=0, help="1: keep the staging barrier of the tile kernels (A/B)")
args = ap.parse_args()

shape = tuple(int(v) for v in args.shape.split(","))
dev = torch.device("cuda:0")
bufs = [torch.randn(*shape, device=dev) for _ in range(3)]
variants = [("single", 2, 0)] + [(f"tile-pair rows={r}", 1, int(r)) for r in args.rows.split(",") if r] + \
    [(f"roll seg={r}", 3, int(r)) for r in args.seg.split(",") if r] + [("auto", 0, 0)]
times = {v[0]: [] for v in variants}
kids = {}


if args.inverse:
    coefs = [ptwt_amd.wavedec2(b, args.wavelet, mode=args.mode, level=args.level) for b in bufs]
    variants = [("single", 2, 0), ("auto", 0, 0)]
    times = {v[0]: [] for v in variants}

    def run(i):
        return ptwt_amd.waverec2(coefs[i % 3], args.wavelet)
else:
    def run(i):
        return ptwt_amd.wavedec2(bufs[i % 3], args.wavelet, mode=args.mode, level=args.level)


for rnd in range(args.rounds + 1):
    for name, pm, rows in variants:
        _engine.set_option(_engine.OPT_PAIR_MODE, pm)
        _engine.set_option(_engine.OPT_PAIR_ROWS, rows)
        _engine.set_option(10, args.sync_stage)
        if rnd == 0:
            _engine.level_events = []
            run(0)
            torch.cuda.synchronize()
            kids[name] = [(e[1], round(e[3].elapsed_time(e[4]) * 1e3, 1)) for e in _engine.level_events]
            _engine.level_events = None
            continue
        for i in range(3):
            run(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(args.iters):
            run(i)
        e1.record()
        torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) * 1e3 / args.iters)
_engine.set_option(_engine.OPT_PAIR_MODE, 0)
_engine.set_option(_engine.OPT_PAIR_ROWS, 0)
_engine.set_option(10, 0)
n = 1
for v in shape:
    n *= v
for name, _, _ in variants:
    t = statistics.median(times[name])
    print(json.dumps({"variant": name, "shape": list(shape), "wavelet": args.wavelet, "level": args.level, "us_per_call": round(t, 1),
                      "min_us": round(min(times[name]), 1), "Msamples_per_s": round(n / t, 0), "levels(kid,us)": kids[name]}))
