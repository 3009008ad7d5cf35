"""A/B timing of the 3-D analysis kernels on BASELINE config 3 (8 x 256^3 db2): slice-per-wave bricks (OPT_TILE_ROWS 0: 2 x 4 x 64,
3: 3 x 4 x 64), row-dealt bricks (2: 2 x 4 x 64, 1: 4 x 4 x 64), composed route (tile mode 2)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g; g.build(verbose=False)
import ptwt_amd
from ptwt_amd import _engine
dev = torch.device("cuda:0")
xs = [torch.randn(8, 256, 256, 256, device=dev) for _ in range(3)]
wav = os.environ.get("MIFWT_WAVELET", "db2")
variants = [("slice 2x4", 0, 0, 0), ("slice 3x4", 0, 3, 0), ("brick 2x4", 0, 2, 0), ("brick 4x4", 0, 1, 0), ("composed", 2, 0, 0)]
for rnd in range(2):
    for name, opt5, opt6, opt1 in variants:
        _engine.set_option(5, opt5)
        _engine.set_option(6, opt6)
        _engine.set_option(_engine.OPT_ROWS_PER_CHUNK, opt1)
        out = []
        for lvl in (1, 3):
            for i in range(3): ptwt_amd.wavedec3(xs[i], wav, level=lvl)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for i in range(12): ptwt_amd.wavedec3(xs[i % 3], wav, level=lvl)
            e.record(); torch.cuda.synchronize()
            out.append(round(s.elapsed_time(e) / 12, 4))
        if rnd:
            print(f"{name:12s} level1 {out[0]} ms   3 levels {out[1]} ms")
_engine.set_option(5, 0); _engine.set_option(6, 0); _engine.set_option(_engine.OPT_ROWS_PER_CHUNK, 0)
