"""Does the power-of-two row / image pitch of config 2 cost the streaming kernels anything?  wavedec2 / waverec2 db4 level 3 on 64 images
of 1024 x W for several W, per call and normalised to 1024 columns."""
import sys, time, torch
sys.path.insert(0, '.')
import ptwt_amd
def loop(fn, n=100):
    for i in range(10): fn(i)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for i in range(n): fn(i)
        torch.cuda.synchronize()
        best = min(best, 1e6 * (time.perf_counter() - t0) / n)
    return best
for W in (1024, 1016, 1008, 1000, 1032, 1040, 1056, 1024):
    xs = [torch.randn(64, 1024, W, device='cuda') for _ in range(3)]
    cs = [ptwt_amd.wavedec2(x, 'db4', level=3) for x in xs]
    a = loop(lambda i: ptwt_amd.wavedec2(xs[i % 3], 'db4', level=3))
    b = loop(lambda i: ptwt_amd.waverec2(cs[i % 3], 'db4'))
    print(f"W = {W}: wavedec2 {a:6.1f} us ({a * 1024 / W:6.1f} per 1024 columns)   waverec2 {b:6.1f} us ({b * 1024 / W:6.1f})", flush=True)
    del xs, cs
