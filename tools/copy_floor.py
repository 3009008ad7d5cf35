#!/usr/bin/env python
"""What does the simplest possible kernel (torch's elementwise copy) need for the byte counts of the pyramid levels of
BASELINE config 2?  Back-to-back launches, rotating buffers — the floor a per-level kernel can be compared with."""
import json, sys, torch
dev = torch.device("cuda:0")
for name, nbytes in [("level3 36MB", 36e6), ("level2 138MB", 138e6), ("level1 540MB", 540e6)]:
    n = int(nbytes / 2 / 4)
    srcs = [torch.randn(n, device=dev) for _ in range(3)]
    dsts = [torch.empty(n, device=dev) for _ in range(3)]
    for i in range(3):
        dsts[i].copy_(srcs[i])
    torch.cuda.synchronize()
    best = 1e9
    for rnd in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(20):
            dsts[i % 3].copy_(srcs[i % 3])
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 20)
    print(json.dumps({"case": name, "ms": round(best, 4), "GBps": round(2 * 4 * n / best / 1e6, 1)}))
