"""Config 3 (waverec3 db2 level 3 on 8 x 256^3): the depth-walking synthesis kernel (tile mode 4) against the bricks; depth segments,
staging depth, A/B switches (debug 1: no stores, 2: no loads, 4: no W / H pass)."""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
from tools.walk3_time import t  # noqa
xs = [torch.randn(8, 256, 256, 256, device='cuda') for _ in range(3)]
cs = [ptwt_amd.wavedec3(x, 'db2', mode='zero', level=3) for x in xs]
c1 = [ptwt_amd.wavedec3(x, 'db2', mode='zero', level=1) for x in xs]
del xs
rec = lambda c: ptwt_amd.waverec3(c, 'db2')
for tm, name in ((1, "bricks"), (4, "walk")):
    _engine.set_option(_engine.OPT_TILE_MODE, tm)
    print(f"{name}: finest level {t(rec, c1):.1f} us, whole waverec3 {t(rec, cs):.1f} us", flush=True)
_engine.set_option(_engine.OPT_TILE_MODE, 4)
for seg, pf in ((0, 2), (0, 3), (0, 4), (0, 5), (128, 2), (128, 4), (64, 2), (64, 4), (16, 2), (16, 4), (8, 2)):
    _engine.set_option(_engine.OPT_ROWS_PER_CHUNK, seg); _engine.set_option(_engine.OPT_PREFETCH_PAIRS, pf)
    print(f"  walk finest level, {seg or 'auto'} slice pairs per segment, {pf} ahead: {t(rec, c1):.1f} us", flush=True)
_engine.set_option(_engine.OPT_ROWS_PER_CHUNK, 0); _engine.set_option(_engine.OPT_PREFETCH_PAIRS, 0)
for dbg, name in ((1, "no stores"), (2, "no loads"), (3, "neither"), (4, "no W / H pass")):
    _engine.set_option(_engine.OPT_DEBUG, dbg)
    print(f"  walk finest level, {name}: {t(rec, c1):.1f} us", flush=True)
_engine.set_option(_engine.OPT_DEBUG, 0)
_engine.set_option(_engine.OPT_NT_STORE, 1)
print(f"  walk finest level, nt stores: {t(rec, c1):.1f} us", flush=True)
_engine.set_option(_engine.OPT_NT_STORE, 0)
_engine.set_option(_engine.OPT_TILE_MODE, 0)
