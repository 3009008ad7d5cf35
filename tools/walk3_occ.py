"""Deep levels of config 3 through the depth-walking analysis kernel: staging depth = workgroups per CU (LDS) against the bricks."""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
from tools.walk3_time import t  # noqa
for n in (129, 66):
    xs = [torch.randn(8, n, n, n, device='cuda') for _ in range(3)]
    f = lambda x: ptwt_amd.wavedec3(x, 'db2', mode='zero', level=1)
    _engine.set_option(_engine.OPT_TILE_MODE, 1)
    print(f"8 x {n}^3 bricks: {t(f, xs, 60):.1f} us", flush=True)
    _engine.set_option(_engine.OPT_TILE_MODE, 4)
    for pf in (1, 2, 3, 4):
        for seg in (0, 6, 12, 17, 33):
            _engine.set_option(_engine.OPT_PREFETCH_PAIRS, pf); _engine.set_option(_engine.OPT_ROWS_PER_CHUNK, seg)
            print(f"8 x {n}^3 walk, {pf} ahead, {seg or 'auto'} slices per segment: {t(f, xs, 60):.1f} us", flush=True)
    _engine.set_option(_engine.OPT_PREFETCH_PAIRS, 0); _engine.set_option(_engine.OPT_ROWS_PER_CHUNK, 0); _engine.set_option(_engine.OPT_TILE_MODE, 0)
