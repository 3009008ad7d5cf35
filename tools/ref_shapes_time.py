"""Whole-call time of the reference's two speed shapes (eager loop, results dropped), us."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptwt_amd
def timeit(fn, n=300):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
x = torch.randn(32, 1000, 1000, device='cuda'); v = torch.randn(32, 100, 100, 100, device='cuda')
import gc; gc.disable()
for rep in range(3):
    print('wavedec2 32 x 1000^2 db5 periodic L5: %.1f us    wavedec3 32 x 100^3 db5 periodic L3: %.1f us' % (
        timeit(lambda: ptwt_amd.wavedec2(x, 'db5', mode='periodic', level=5)), timeit(lambda: ptwt_amd.wavedec3(v, 'db5', mode='periodic', level=3))))
