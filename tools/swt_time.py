"""swt / iswt (1-D stationary transform) and 2-D packets on big inputs: time against the bytes they must move."""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    r = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); r.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(r)[2]
for shape, wav, lev in (((64, 1 << 20), 'db4', 3), ((4096, 4096), 'db2', 4), ((32, 1000000), 'db5', 5)):
    x = torch.randn(*shape, device='cuda')
    n = x.numel() * 4
    f = lambda: ptwt_amd.swt(x, wav, level=lev)
    us = t(f)
    byt = n * (1 + lev) + n * (2 * lev - 1 + 1)  # every level: read the approximation, write approximation + details; all kept
    print(f'swt {wav} level {lev} on {shape}: {us:.1f} us; per-level bytes {3 * lev * n / 1e6:.0f} MB -> {3 * lev * n / us / 8e6:.3f} of 8 TB/s', flush=True)
    c = f()
    g = lambda: ptwt_amd.iswt(c, wav)
    us = t(g)
    print(f'iswt: {us:.1f} us -> {3 * lev * n / us / 8e6:.3f}', flush=True)
    del c
x = torch.randn(16, 1024, 1024, device='cuda')
for lev in (2, 3):
    f = lambda: ptwt_amd.WaveletPacket2D(x, 'db4', mode='reflect', maxlevel=lev)
    us = t(f, 5)
    nb = x.numel() * 4
    print(f'WaveletPacket2D db4 maxlevel {lev} on 16 x 1024^2: {us:.1f} us; {2 * lev * nb / 1e6:.0f} MB read + written over the levels -> {2 * lev * nb / us / 8e6:.3f} of 8 TB/s', flush=True)
