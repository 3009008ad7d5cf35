"""waverec3 of the reference's 3-D speed shape (32 x 100^3 db5 periodic level 3): 60 calls (for a kernel trace)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptwt_amd
wav = sys.argv[1] if len(sys.argv) > 1 else 'db5'
tm = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = int(sys.argv[3]) if len(sys.argv) > 3 else 100
x = torch.randn(32, n, n, n, device='cuda')
c = ptwt_amd.wavedec3(x, wav, mode='periodic', level=3)
from ptwt_amd import _engine as E
E.set_option(E.OPT_TILE_MODE, tm)
for _ in range(60): ptwt_amd.waverec3(c, wav)
torch.cuda.synchronize()
