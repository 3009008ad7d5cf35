"""wavedec3 / waverec3 db2 level 3 on 8 x 256^3 f32, whole calls: auto routing against the walk kernels on every level (MIFWT_OPT_TILE_MODE 4)."""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    r = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); r.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(r)[3]
xs = [torch.randn(8, 256, 256, 256, device='cuda') for _ in range(3)]
i = [0]
def f():
    i[0] += 1; return ptwt_amd.wavedec3(xs[i[0] % 3], 'db2', mode='zero', level=3)
c = f()
g = lambda: ptwt_amd.waverec3(c, 'db2')
for rep in range(3):
    for tm in (0, 4):
        _engine.set_option(_engine.OPT_TILE_MODE, tm)
        _engine.level_events = []; f(); g(); kids = [e[1] for e in _engine.level_events]; _engine.level_events = None
        print(f'tile mode {tm}: kernels {kids}; wavedec3 {t(f):.1f} us, waverec3 {t(g):.1f} us', flush=True)
_engine.set_option(_engine.OPT_TILE_MODE, 0)
