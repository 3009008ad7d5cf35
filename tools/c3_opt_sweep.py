"""Config 3 (wavedec3 / waverec3 db2 level 3 on 8 x 256^3) against the library's A/B options: non-temporal band stores, brick rows,
the composed route."""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
xs = [torch.randn(8, 256, 256, 256, device='cuda') for _ in range(3)]
cs = [ptwt_amd.wavedec3(x, 'db2', level=3) for x in xs]
def t(fn, args, kw):
    for i in range(5): fn(args[i % 3], 'db2', **kw)
    torch.cuda.synchronize()
    r = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(30): fn(args[i % 3], 'db2', **kw)
        e1.record(); torch.cuda.synchronize(); r.append(e0.elapsed_time(e1) / 30)
    return sorted(r)[2]
for name, opts in (("default", {}), ("nt stores", {_engine.OPT_NT_STORE: 1}), ("tile rows 2", {_engine.OPT_TILE_ROWS: 2}), ("tile rows 8", {_engine.OPT_TILE_ROWS: 8}),
                   ("composed route", {_engine.OPT_TILE_MODE: 2})):
    for k, v in opts.items(): _engine.set_option(k, v)
    try:
        print(f"{name}: wavedec3 {t(ptwt_amd.wavedec3, xs, dict(level=3)):.4f} ms   waverec3 {t(ptwt_amd.waverec3, cs, {}):.4f} ms", flush=True)
    finally:
        for k in opts: _engine.set_option(k, 0)
