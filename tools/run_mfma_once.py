import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g; g.build(verbose=False)
import ptwt_amd
ptwt_amd.set_half_storage(True)
x = [torch.randn(8, 8192, 8192, device="cuda:0").half() for _ in range(2)]
for i in range(4):
    ptwt_amd.wavedec2(x[i % 2], "sym16", level=1)
torch.cuda.synchronize()
