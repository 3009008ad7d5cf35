"""The reference's published shapes (db5, periodic): output rows per tile of the LDS-tile analysis kernel (MIFWT_OPT_TILE_ROWS)."""
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
x2 = torch.randn(32, 1000, 1000, device='cuda')
x3 = torch.randn(32, 100, 100, 100, device='cuda')
x4 = torch.randn(64, 4096, 4096, device='cuda')
f2 = lambda: ptwt_amd.wavedec2(x2, 'db5', mode='periodic', level=5)
f3 = lambda: ptwt_amd.wavedec3(x3, 'db5', mode='periodic', level=3)
f4 = lambda: ptwt_amd.wavedec2(x4, 'db8', mode='reflect', level=4)
for rep in range(2):
    for rows in (0, 8, 12, 16, 20, 24):
        _engine.set_option(_engine.OPT_TILE_ROWS, rows)
        print(f'rows per tile {rows or "default"}: wavedec2 32 x 1000^2 db5 L5 {t(f2):.1f} us, wavedec3 32 x 100^3 db5 L3 {t(f3):.1f} us, wavedec2 64 x 4096^2 db8 L4 {t(f4, 4):.0f} us', flush=True)
_engine.set_option(_engine.OPT_TILE_ROWS, 0)
