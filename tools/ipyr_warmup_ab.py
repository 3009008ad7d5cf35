"""A/B of kernel 22's fast warm-up (MIFWT_OPT_DEBUG bit 8 switches it off): waverec2 per call on a few shapes, alternating rounds."""
import sys, time, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine

SHAPES = [((64, 1024, 1024), 'db4', 3, 'reflect'), ((16, 1024, 1024), 'db4', 3, 'reflect'), ((4, 1024, 1024), 'db4', 3, 'reflect'),
          ((32, 1000, 1000), 'db5', 5, 'periodic'), ((256, 384, 384), 'db4', 3, 'reflect'), ((64, 1024, 1024), 'haar', 3, 'reflect'),
          ((64, 1024, 1024), 'db2', 2, 'reflect'), ((8, 1500, 1500), 'db2', 3, 'reflect')]
for shape, wav, lev, mode in SHAPES:
    xs = [torch.randn(*shape, device='cuda') for _ in range(3)]
    cs = [ptwt_amd.wavedec2(x, wav, level=lev, mode=mode) for x in xs]
    res = {0: [], 8: []}
    for rnd in range(3):
        for dbg in (0, 8):
            _engine.set_option(_engine.OPT_DEBUG, dbg)
            for i in range(20): ptwt_amd.waverec2(cs[i % 3], wav)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(200): ptwt_amd.waverec2(cs[i % 3], wav)
            torch.cuda.synchronize()
            res[dbg].append(1e6 * (time.perf_counter() - t0) / 200)
    _engine.set_option(_engine.OPT_DEBUG, 0)
    print(f"{shape} {wav} L{lev} {mode}: fast warm-up {min(res[0]):6.1f} us (rounds {' '.join(f'{v:.1f}' for v in res[0])}); off {min(res[8]):6.1f} us ({' '.join(f'{v:.1f}' for v in res[8])})", flush=True)
