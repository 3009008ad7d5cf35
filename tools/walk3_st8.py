"""Config 3 level 1 through the depth-walking analysis kernel: 8-byte band stores (lane-pair exchange of a row pair) against dword
stores (MIFWT_OPT_DEBUG 512)."""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
from tools.walk3_time import t  # noqa
xs = [torch.randn(8, 256, 256, 256, device='cuda') for _ in range(3)]
f1 = lambda x: ptwt_amd.wavedec3(x, 'db2', mode='zero', level=1)
f3 = lambda x: ptwt_amd.wavedec3(x, 'db2', mode='zero', level=3)
for rep in range(3):
    for dbg, name in ((0, "8-byte stores"), (512, "dword stores"), (2, "8-byte stores, no loads"), (512 + 2, "dword stores, no loads")):
        _engine.set_option(_engine.OPT_DEBUG, dbg)
        print(f"{name}: level 1 {t(f1, xs):.1f} us, whole wavedec3 {t(f3, xs):.1f} us", flush=True)
_engine.set_option(_engine.OPT_DEBUG, 0)
