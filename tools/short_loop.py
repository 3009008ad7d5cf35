"""What a K = 20 timed loop (the driver's command) pays on top of the steady-state time per step: the host time of the first call
(the GPU idles until it is enqueued), the wake-up of the final synchronisation.  Config 2, wavedec2 db4 level 3."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptwt_amd
dev = torch.device('cuda:0')
xs = [torch.randn(64, 1024, 1024, device=dev) for _ in range(3)]
f = lambda i: ptwt_amd.wavedec2(xs[i % 3], 'db4', mode='reflect', level=3)
for i in range(300): f(i)
torch.cuda.synchronize()
import gc; gc.disable()
def loop(k):
    for i in range(5): f(i)
    torch.cuda.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(k): f(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e6
r20 = sorted(loop(20) for _ in range(15)); r200 = sorted(loop(200) for _ in range(5))
# host time of one call when the queue is not the bottleneck
torch.cuda.synchronize(); t0 = time.perf_counter(); f(0); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("HSA_ENABLE_INTERRUPT=%s  K=20: median %.2f min %.2f us/step   K=200: median %.2f   first call enqueue %.1f us, call + sync %.1f us" % (
    os.environ.get("HSA_ENABLE_INTERRUPT"), r20[7], r20[0], r200[2], (t1 - t0) * 1e6, (t2 - t0) * 1e6))
