"""Batch sweep of wavedec2 db4 level 3 on 1024 x 1024 planes: time per call against the linear trend through B = 64."""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
def t(fn, n=40):
    for _ in range(8): fn()
    torch.cuda.synchronize()
    r = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); r.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(r)[2]
res = {}
rot = {}
Bs = [int(v) for v in sys.argv[1].split(',')] if len(sys.argv) > 1 else [16, 32, 48, 64, 65, 72, 80, 96, 100, 128, 192, 256]
for B in Bs:
    xs = [torch.randn(B, 1024, 1024, device='cuda') for _ in range(3)]
    i = [0]
    def f():
        i[0] += 1; return ptwt_amd.wavedec2(xs[i[0] % 3], 'db4', level=3)
    _engine.level_events = []
    f(); torch.cuda.synchronize()
    kids = [e[1] for e in _engine.level_events]; _engine.level_events = None
    res[B] = t(f)
    held = [None, None, None]
    def frot():
        i[0] += 1; held[i[0] % 3] = ptwt_amd.wavedec2(xs[i[0] % 3], 'db4', level=3)
    rot[B] = t(frot)
    held[:] = [None, None, None]
    byts = 4 * B * (1024 * 1024 + 3 * 515 * 515 + 3 * 261 * 261 + 4 * 134 * 134)
    print(f'B={B:4d} kernels {kids}: results dropped {res[B]:7.1f} us = {byts / res[B] / 8e6:.3f} of 8 TB/s; rotating output sets {rot[B]:7.1f} us = {byts / rot[B] / 8e6:.3f}', flush=True)
    del xs, held; torch.cuda.empty_cache()
if 64 in res:
    print(f'against the lines through B = 64 ({res[64]:.1f} us with the results dropped: 64 images x 4.3 MB of output just about fit the 256 MiB Infinity Cache, larger batches do not; {rot[64]:.1f} us with rotating output sets: every byte to HBM at every batch size):')
    for B in Bs: print(f'  B={B:4d}: results dropped {res[B] / (res[64] * B / 64):.3f}   rotating {rot[B] / (rot[64] * B / 64):.3f}   results dropped against the ROTATING line {res[B] / (rot[64] * B / 64):.3f}')
