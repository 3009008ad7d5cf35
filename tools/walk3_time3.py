"""The deep levels of config 3 on their own (8 x 129^3 and 8 x 66^3 volumes, db2, zero mode, one level): depth-walking kernel against
the bricks, depth segments / staging depth / request policy."""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
from tools.walk3_time import t  # noqa
for n in (129, 128, 66, 64):
    xs = [torch.randn(8, n, n, n, device='cuda') for _ in range(3)]
    f = lambda x: ptwt_amd.wavedec3(x, 'db2', mode='zero', level=1)
    _engine.set_option(_engine.OPT_TILE_MODE, 1)
    print(f"8 x {n}^3 bricks: {t(f, xs, 60):.1f} us", flush=True)
    _engine.set_option(_engine.OPT_TILE_MODE, 4)
    for dbg in (0, 32):
        for seg, pf in ((0, 0), (0, 4), (0, 6), (4, 4), (4, 6), (6, 4), (6, 6), (16, 4), (33, 4)):
            _engine.set_option(_engine.OPT_DEBUG, dbg); _engine.set_option(_engine.OPT_ROWS_PER_CHUNK, seg); _engine.set_option(_engine.OPT_PREFETCH_PAIRS, pf)
            print(f"8 x {n}^3 walk, {'default policy' if dbg else 'non-temporal'}, {seg or 'auto'} slices per segment, {pf or 3} ahead: {t(f, xs, 60):.1f} us", flush=True)
    _engine.set_option(_engine.OPT_DEBUG, 0); _engine.set_option(_engine.OPT_ROWS_PER_CHUNK, 0); _engine.set_option(_engine.OPT_PREFETCH_PAIRS, 0)
    _engine.set_option(_engine.OPT_TILE_MODE, 0)
