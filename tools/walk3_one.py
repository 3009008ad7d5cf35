"""One level of a 3-D decomposition of 32 volumes, 60 calls (for a kernel trace).  usage: walk3_one.py <wavelet> <extent> <tile mode> [batch]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptwt_amd as ptwt
from ptwt_amd import _engine as E
wav, n, tm = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
b = int(sys.argv[4]) if len(sys.argv) > 4 else 32
x = torch.randn(b, n, n, n, device='cuda')
E.set_option(E.OPT_TILE_MODE, tm)
for _ in range(60): ptwt.wavedec3(x, wav, mode=os.environ.get('W3MODE', 'periodic'), level=1)
torch.cuda.synchronize()
