"""cProfile of the host side of ptwt_amd.wavedec2 (config 2) — where the ~25 us of Python per call go."""
import cProfile, os, pstats, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptwt_amd
x = torch.randn(64, 1024, 1024, device='cuda')
f = lambda: ptwt_amd.wavedec2(x, 'db4', mode='reflect', level=3)
for _ in range(200): f()
torch.cuda.synchronize()
import gc; gc.disable()
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): f()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats('tottime'); st.print_stats(28)
