"""Config 3 level 1 through the depth-walking analysis kernel, current defaults: how much of the launch is arithmetic (debug 4: no W / H
pass; 1 / 2: no stores / no loads)."""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
from tools.walk3_time import t  # noqa
xs = [torch.randn(8, 256, 256, 256, device='cuda') for _ in range(3)]
f = lambda x: ptwt_amd.wavedec3(x, 'db2', mode='zero', level=1)
for rep in range(2):
    for dbg, name in ((0, "as shipped"), (4, "no W / H pass"), (5, "no W / H pass, no stores"), (6, "no W / H pass, no loads"), (7, "barriers + D pass only"), (1, "no stores"), (2, "no loads"), (3, "neither")):
        _engine.set_option(_engine.OPT_DEBUG, dbg)
        print(f"{name}: {t(f, xs):.1f} us", flush=True)
_engine.set_option(_engine.OPT_DEBUG, 0)
