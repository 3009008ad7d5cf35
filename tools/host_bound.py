"""Is a call loop host-bound?  200 calls enqueued without a sync: time until the last call RETURNS (host) vs until the GPU is done."""
import sys, time, torch
sys.path.insert(0, '.')
import ptwt_amd
xs = [torch.randn(64, 1024, 1024, device='cuda') for _ in range(3)]
cs = [ptwt_amd.wavedec2(x, 'db4', level=3) for x in xs]
for name, call in (("wavedec2", lambda i: ptwt_amd.wavedec2(xs[i % 3], 'db4', level=3)), ("waverec2", lambda i: ptwt_amd.waverec2(cs[i % 3], 'db4'))):
    for i in range(30): call(i)
    torch.cuda.synchronize()
    for rnd in range(3):
        t0 = time.perf_counter()
        for i in range(200): call(i)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{name}: host enqueue {1e6 * (t1 - t0) / 200:6.1f} us/call, until the GPU is done {1e6 * (t2 - t0) / 200:6.1f} us/call")
