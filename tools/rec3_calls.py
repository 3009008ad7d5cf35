"""waverec3 of the reference's 3-D speed shape (32 x 100^3 db5 periodic level 3): 60 calls (for a kernel trace)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptwt_amd
wav = sys.argv[1] if len(sys.argv) > 1 else 'db5'
x = torch.randn(32, 100, 100, 100, device='cuda')
c = ptwt_amd.wavedec3(x, wav, mode='periodic', level=3)
for _ in range(60): ptwt_amd.waverec3(c, wav)
torch.cuda.synchronize()
