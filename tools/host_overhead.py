#!/usr/bin/env python
"""How long does the host need to ENQUEUE one wavedec2 call (no sync) vs. how long the GPU needs to run it?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as entry
entry.build(verbose=False)
import ptwt_amd
dev = torch.device("cuda:0")
for shape in [(64, 1024, 1024), (8, 128, 128), (4096, 64, 64), (256, 256, 256)]:
    xs = [torch.randn(*shape, device=dev) for _ in range(3)]
    for i in range(5):
        ptwt_amd.wavedec2(xs[i % 3], "db4", level=3)
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for i in range(n):
        ptwt_amd.wavedec2(xs[i % 3], "db4", level=3)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"shape {shape}: enqueue {1e6*(t1-t0)/n:.1f} us/call, total {1e6*(t2-t0)/n:.1f} us/call")
import cProfile, pstats
x = torch.randn(8, 64, 64, device=dev)
pr = cProfile.Profile()
pr.enable()
for i in range(300):
    ptwt_amd.wavedec2(x, "db4", level=3)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(32)
