"""Kernel 16 on column-strided views of one 64 x 1024 x 1024 buffer (row pitch stays 4096 bytes): what do the nearly empty last waves
of each level cost?  W = 1024 runs 5 + 3 + 3 level waves, W <= 975 runs 4 + 2 + 2."""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
def t(fn, n=50):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    r = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); r.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(r)[2]
B = 64
big = [torch.randn(B, 1024, 1024, device='cuda') for _ in range(3)]
for wav in ('db4', 'db2', 'haar'):
    L = len(ptwt_amd._wavelets.host_taps(wav)[0])
    for H, W in ((1024, 1024), (1024, 1008), (1024, 992), (1024, 976), (1024, 968), (1024, 960), (1024, 896), (976, 976), (1018, 1018)):
        xs = [b[:, :H, :W] for b in big]
        i = [0]
        def f():
            i[0] += 1; return ptwt_amd.wavedec2(xs[i[0] % 3], wav, level=3)
        _engine.level_events = []
        c = f(); torch.cuda.synchronize()
        kids = [e[1] for e in _engine.level_events]; _engine.level_events = None
        us = t(f)
        byts = 4 * (B * H * W + c[0].numel() + sum(v.numel() for d in c[1:] for v in d))
        w = [W]
        for _ in range(3): w.append((w[-1] + L - 1) // 2)
        print(f'{wav} {H}x{W} kernels {kids} widths {w[1:]}: {us:.1f} us, {byts / 1e6:.0f} MB, {byts / us / 1e6:.2f} TB/s = {byts / us / 8e6:.3f}', flush=True)
