"""f64 wavedec2 / waverec2 on config 2 (64 x 1024^2 db4 level 3): output rows per tile of the LDS-tile kernels (MIFWT_OPT_TILE_ROWS)."""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    r = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); r.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(r)[2]
xs = [torch.randn(64, 1024, 1024, device='cuda', dtype=torch.float64) for _ in range(3)]
i = [0]
def f():
    i[0] += 1; return ptwt_amd.wavedec2(xs[i[0] % 3], 'db4', level=3)
c = f()
def g():
    return ptwt_amd.waverec2(c, 'db4')
for rep in range(2):
    for rows in (0, 8, 12, 16, 20, 24):
        _engine.set_option(_engine.OPT_TILE_ROWS, rows)
        print(f'rows per tile {rows or "default"}: wavedec2 {t(f):.1f} us, waverec2 {t(g):.1f} us', flush=True)
_engine.set_option(_engine.OPT_TILE_ROWS, 0)
