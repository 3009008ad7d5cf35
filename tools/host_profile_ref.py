"""Host side of the reference's speed shapes (many small launches per call): wall time per call when nothing waits for the GPU (a tiny
batch: every kernel is over before the next launch), and the cProfile of it.  usage: host_profile_ref.py 2d|3d"""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptwt_amd
which = sys.argv[1] if len(sys.argv) > 1 else '2d'
if which == '2d':
    x = torch.randn(1, 1000, 1000, device='cuda')
    f = lambda: ptwt_amd.wavedec2(x, 'db5', mode='periodic', level=5)
else:
    x = torch.randn(1, 100, 100, 100, device='cuda')
    f = lambda: ptwt_amd.wavedec3(x, 'db5', mode='periodic', level=3)
for _ in range(300): f()
torch.cuda.synchronize()
import gc; gc.disable()
t0 = time.perf_counter()
for _ in range(2000): f()
t1 = time.perf_counter(); torch.cuda.synchronize()
print('%s: host %.1f us per call' % (which, (t1 - t0) / 2000 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): f()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats('tottime'); st.print_stats(32)
