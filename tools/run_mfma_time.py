import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g; g.build(verbose=False)
import ptwt_amd
from ptwt_amd import _engine
ptwt_amd.set_half_storage(True)
dev = torch.device("cuda:0")
x = [torch.randn(8, 8192, 8192, device=dev).half() for _ in range(3)]
for name, opt in (("mfma", 0), ("vector", 2), ("mfma", 0)):
    _engine.set_option(7, opt)
    for i in range(3): ptwt_amd.wavedec2(x[i], "sym16", level=1)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(9): ptwt_amd.wavedec2(x[i % 3], "sym16", level=1)
    e.record(); torch.cuda.synchronize()
    print(name, round(s.elapsed_time(e) / 9, 4), "ms")
