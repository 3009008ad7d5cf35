"""Per-call GPU time of the first calls after a synchronize (does the first timed step of bench.py pay a ramp?)."""
import sys, time, torch
sys.path.insert(0, '.')
import ptwt_amd
xs = [torch.randn(64, 1024, 1024, device='cuda') for _ in range(3)]
t0 = time.perf_counter()
i = 0
while time.perf_counter() - t0 < 0.03:
    ptwt_amd.wavedec2(xs[i % 3], 'db4', level=3); i += 1
torch.cuda.synchronize()
for rep in range(3):
    for i in range(5): ptwt_amd.wavedec2(xs[i % 3], 'db4', level=3)
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(20):
        ptwt_amd.wavedec2(xs[i % 3], 'db4', level=3)
        evs[i + 1].record()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print('wall per step %.1f us; per-call GPU us:' % ((t1 - t0) / 20 * 1e6), ' '.join('%.0f' % (evs[k].elapsed_time(evs[k + 1]) * 1e3) for k in range(20)))
    t0 = time.perf_counter()
    for i in range(20): ptwt_amd.wavedec2(xs[i % 3], 'db4', level=3)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print('  without events: wall per step %.1f us' % ((t1 - t0) / 20 * 1e6))
