"""Config 3, both directions: whole-call GPU time and host enqueue time per call."""
import sys, time, torch
sys.path.insert(0, '.')
import ptwt_amd
from tools.walk3_time import t  # noqa
xs = [torch.randn(8, 256, 256, 256, device='cuda') for _ in range(3)]
cs = [ptwt_amd.wavedec3(x, 'db2', mode='zero', level=3) for x in xs]
for name, fn, args in (("wavedec3", lambda x: ptwt_amd.wavedec3(x, 'db2', mode='zero', level=3), xs), ("waverec3", lambda c: ptwt_amd.waverec3(c, 'db2'), cs)):
    g = t(fn, args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(50): fn(args[i % 3])
    h = (time.perf_counter() - t0) / 50 * 1e6
    torch.cuda.synchronize()
    print(f"{name}: {g:.1f} us per call on the GPU, {h:.1f} us of host time to enqueue one", flush=True)
