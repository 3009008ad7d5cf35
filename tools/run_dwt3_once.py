import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g; g.build(verbose=False)
import ptwt_amd
x = [torch.randn(8, 256, 256, 256, device="cuda:0") for _ in range(2)]
for i in range(4):
    ptwt_amd.wavedec3(x[i % 2], "db2", level=1)
torch.cuda.synchronize()
