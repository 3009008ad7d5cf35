"""Config 3, both directions: per-level GPU times (level_events) of the default routes and of the depth-walking kernels (tile mode 4)."""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
from tools.walk3_time import t  # noqa
xs = [torch.randn(8, 256, 256, 256, device='cuda') for _ in range(2)]
cs = [ptwt_amd.wavedec3(x, 'db2', mode='zero', level=3) for x in xs]
def levels(fn, arg):
    for _ in range(3): fn(arg)
    torch.cuda.synchronize()
    acc = {}
    for _ in range(10):
        _engine.level_events = []
        fn(arg)
        torch.cuda.synchronize()
        for e in _engine.level_events:
            acc.setdefault((e[1], e[2]), []).append(e[3].elapsed_time(e[4]) * 1e3)
        _engine.level_events = None
    return {k: round(sorted(v)[len(v) // 2], 1) for k, v in acc.items()}
for tm, name in ((0, "auto"), (1, "bricks"), (4, "walk")):
    _engine.set_option(_engine.OPT_TILE_MODE, tm)
    print(name, "wavedec3", levels(lambda x: ptwt_amd.wavedec3(x, 'db2', mode='zero', level=3), xs[0]), f"whole {t(lambda x: ptwt_amd.wavedec3(x, 'db2', mode='zero', level=3), xs):.1f} us", flush=True)
    print(name, "waverec3", levels(lambda c: ptwt_amd.waverec3(c, 'db2'), cs[0]), f"whole {t(lambda c: ptwt_amd.waverec3(c, 'db2'), cs):.1f} us", flush=True)
_engine.set_option(_engine.OPT_TILE_MODE, 4)
for seg in (4, 5, 6, 8, 11, 16):
    _engine.set_option(_engine.OPT_ROWS_PER_CHUNK, seg)
    print(f"walk, {seg} per segment: waverec3", levels(lambda c: ptwt_amd.waverec3(c, 'db2'), cs[0]), flush=True)
_engine.set_option(_engine.OPT_ROWS_PER_CHUNK, 0)
for pf in (1, 2, 3):
    _engine.set_option(_engine.OPT_PREFETCH_PAIRS, pf)
    print(f"walk, {pf} ahead: waverec3", levels(lambda c: ptwt_amd.waverec3(c, 'db2'), cs[0]), flush=True)
_engine.set_option(_engine.OPT_PREFETCH_PAIRS, 0)
_engine.set_option(_engine.OPT_TILE_MODE, 0)
