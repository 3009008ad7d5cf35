"""Config 3 finest synthesis level through the depth-walking kernel: 16-byte output stores (lane-pair exchange) against 8-byte stores
(MIFWT_OPT_DEBUG 512)."""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
from tools.walk3_time import t  # noqa
xs = [torch.randn(8, 256, 256, 256, device='cuda') for _ in range(3)]
c1 = [ptwt_amd.wavedec3(x, 'db2', mode='zero', level=1) for x in xs]
cs = [ptwt_amd.wavedec3(x, 'db2', mode='zero', level=3) for x in xs]
del xs
rec = lambda c: ptwt_amd.waverec3(c, 'db2')
for rep in range(3):
    for dbg, name in ((0, "16-byte stores"), (512, "8-byte stores"), (2, "16-byte stores, no loads"), (512 + 2, "8-byte stores, no loads")):
        _engine.set_option(_engine.OPT_DEBUG, dbg)
        print(f"{name}: finest level {t(rec, c1):.1f} us, whole waverec3 {t(rec, cs):.1f} us", flush=True)
_engine.set_option(_engine.OPT_DEBUG, 0)
