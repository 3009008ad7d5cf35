import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g; g.build(verbose=False)
import ptwt_amd
from ptwt_amd import _engine
dev = torch.device("cuda:0")
xs = [torch.randn(8, 256, 256, 256, device=dev) for _ in range(3)]
for name, opt5 in (("brick", 0), ("composed", 2), ("brick", 0)):
    _engine.set_option(5, opt5)
    for lvl in (1, 3):
        for i in range(3): ptwt_amd.wavedec3(xs[i], "db2", level=lvl)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(12): ptwt_amd.wavedec3(xs[i % 3], "db2", level=lvl)
        e.record(); torch.cuda.synchronize()
        print(name, "levels", lvl, round(s.elapsed_time(e) / 12, 4), "ms")
_engine.set_option(5, 0)
