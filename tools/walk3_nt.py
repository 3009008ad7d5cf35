"""Config 3 level 1 through the depth-walking analysis kernel: non-temporal band stores (MIFWT_OPT_NT_STORE) against the default policy."""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
from tools.walk3_time import t  # noqa
xs = [torch.randn(8, 256, 256, 256, device='cuda') for _ in range(3)]
f = lambda x: ptwt_amd.wavedec3(x, 'db2', mode='zero', level=1)
for rep in range(3):
    for nt in (0, 1):
        _engine.set_option(_engine.OPT_NT_STORE, nt)
        a = t(f, xs)
        _engine.set_option(_engine.OPT_DEBUG, 2)
        b = t(f, xs)
        _engine.set_option(_engine.OPT_DEBUG, 0)
        print(f"nt stores {nt}: level 1 {a:.1f} us, without loads {b:.1f} us", flush=True)
_engine.set_option(_engine.OPT_NT_STORE, 0)
