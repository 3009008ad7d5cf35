"""Config 3 (8 x 256^3 db2 level 3, zero mode as bench.py runs it): 60 wavedec3 calls and 60 waverec3 calls (for a kernel trace)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptwt_amd
mode = sys.argv[1] if len(sys.argv) > 1 else 'zero'
x = torch.randn(8, 256, 256, 256, device='cuda')
for _ in range(60): c = ptwt_amd.wavedec3(x, 'db2', mode=mode, level=3)
torch.cuda.synchronize()
for _ in range(60): ptwt_amd.waverec3(c, 'db2')
torch.cuda.synchronize()
