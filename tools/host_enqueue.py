"""Host time per call (enqueue only: the loop is timed up to the point where the last call returns, the GPU far behind) against the
synchronised time per call, for the config-2 calls."""
import gc, sys, time, torch
sys.path.insert(0, '.')
import ptwt_amd
xs = [torch.randn(64, 1024, 1024, device='cuda') for _ in range(3)]
cs = [ptwt_amd.wavedec2(x, 'db4', level=3) for x in xs]
gc.disable()
for name, fn in (('wavedec2', lambda i: ptwt_amd.wavedec2(xs[i % 3], 'db4', level=3)), ('waverec2', lambda i: ptwt_amd.waverec2(cs[i % 3], 'db4'))):
    for i in range(20): fn(i)
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for i in range(n): fn(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'{name}: enqueue {1e6 * (t1 - t0) / n:.1f} us per call, synchronised {1e6 * (t2 - t0) / n:.1f} us per call')
