"""The reference's two published speed shapes, 100 whole calls each after a spin-up (for a kernel trace: tools/ktrace.sh)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptwt_amd
which = sys.argv[1] if len(sys.argv) > 1 else 'both'
if which in ('2d', 'both'):
    x = torch.randn(32, 1000, 1000, device='cuda')
    for i in range(150): ptwt_amd.wavedec2(x, 'db5', mode='periodic', level=5)
    torch.cuda.synchronize()
if which in ('3d', 'both'):
    v = torch.randn(32, 100, 100, 100, device='cuda')
    for i in range(150): ptwt_amd.wavedec3(v, 'db5', mode='periodic', level=3)
    torch.cuda.synchronize()
