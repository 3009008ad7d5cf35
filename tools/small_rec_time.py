"""waverec2 / wavedec2 of small planes: per-call time in call loops (host enqueue vs until the GPU is done)."""
import sys, time, gc, torch
sys.path.insert(0, '.')
import ptwt_amd
xs = [torch.randn(4096, 64, 64, device='cuda') for _ in range(3)]
cs = [ptwt_amd.wavedec2(x, 'db2', level=3) for x in xs]
for name, call in (("wavedec2", lambda i: ptwt_amd.wavedec2(xs[i % 3], 'db2', level=3)), ("waverec2", lambda i: ptwt_amd.waverec2(cs[i % 3], 'db2'))):
    for i in range(30): call(i)
    torch.cuda.synchronize()
    for rnd in range(4):
        g0 = gc.get_count()
        t0 = time.perf_counter()
        for i in range(200): call(i)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{name}: host enqueue {1e6 * (t1 - t0) / 200:6.1f} us/call, until the GPU is done {1e6 * (t2 - t0) / 200:6.1f} us/call   gc counts before {g0} after {gc.get_count()}")
