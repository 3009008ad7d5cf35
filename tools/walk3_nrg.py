"""Row sub-groups per workgroup of the depth-walking analysis kernel (MIFWT_OPT_PAIR_ROWS 1 / 2 / 4): the reference's 3-D shape
(32 x 100^3 db5 periodic), config 3's levels, 16 x 128^3 db4."""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
from tools.walk3_time import t  # noqa
def sweep(label, shape, wavelet, mode, nrgs, extra=()):
    xs = [torch.randn(*shape, device='cuda') for _ in range(3)]
    f = lambda x: ptwt_amd.wavedec3(x, wavelet, mode=mode, level=1)
    _engine.set_option(_engine.OPT_TILE_MODE, 1)
    print(f"{label}: bricks / composed {t(f, xs, 40):.1f} us", flush=True)
    _engine.set_option(_engine.OPT_TILE_MODE, 4)
    for nrg in nrgs:
        for pf in (0,) + tuple(extra):
            _engine.set_option(_engine.OPT_PAIR_ROWS, nrg); _engine.set_option(_engine.OPT_PREFETCH_PAIRS, pf)
            print(f"{label}: walk, {nrg or 'auto'} row sub-groups, {pf or 4} ahead: {t(f, xs, 40):.1f} us", flush=True)
    _engine.set_option(_engine.OPT_PAIR_ROWS, 0); _engine.set_option(_engine.OPT_PREFETCH_PAIRS, 0); _engine.set_option(_engine.OPT_TILE_MODE, 0)
sweep("32 x 100^3 db5 periodic", (32, 100, 100, 100), "db5", "periodic", (0, 1, 2, 4), (2, 3))
sweep("32 x 54^3 db5 periodic", (32, 54, 54, 54), "db5", "periodic", (0, 1, 2, 4))
sweep("8 x 256^3 db2 zero", (8, 256, 256, 256), "db2", "zero", (1, 2), (3,))
sweep("8 x 129^3 db2 zero", (8, 129, 129, 129), "db2", "zero", (1, 2))
sweep("8 x 66^3 db2 zero", (8, 66, 66, 66), "db2", "zero", (1, 2, 4))
sweep("16 x 128^3 db4 reflect", (16, 128, 128, 128), "db4", "reflect", (1, 2, 4))
