"""Config 3 level 1 through the depth-walking analysis kernel: A/B switches (debug 8: rows stored on a 128-sample pitch — wrong results,
line-aligned stores; 64: strips of 64 columns; 1 / 2: no stores / no loads)."""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
from tools.walk3_time import t  # noqa
xs = [torch.randn(8, 256, 256, 256, device='cuda') for _ in range(3)]
f = lambda x: ptwt_amd.wavedec3(x, 'db2', mode='zero', level=1)
_engine.set_option(_engine.OPT_TILE_MODE, 4)
for rep in range(2):
    for dbg, name in ((0, "as shipped"), (8, "line-aligned rows (wrong results)"), (8 + 64, "line-aligned rows, 64-column strips"), (64, "64-column strips"), (8 + 64 + 2, "aligned, 64-column strips, no loads"), (2, "no loads")):
        _engine.set_option(_engine.OPT_DEBUG, dbg)
        print(f"{name}: {t(f, xs):.1f} us", flush=True)
_engine.set_option(_engine.OPT_DEBUG, 0)
