"""Config 4 slice (db8 level 4 on 64x4096^2), both directions, against MIFWT_OPT_ROWS_PER_CHUNK (output rows per task of the streaming
wave-strip kernels, ids 1 / 2) and the tile kernels instead (MIFWT_OPT_TILE_MODE 1)."""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
xs = [torch.randn(64, 4096, 4096, device='cuda') for _ in range(2)]
cs = [ptwt_amd.wavedec2(x, 'db8', level=4) for x in xs]
def t(fn, args, kw):
    for i in range(3): fn(args[i % 2], 'db8', **kw)
    torch.cuda.synchronize()
    best = []
    for r in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(8): fn(args[i % 2], 'db8', **kw)
        e1.record(); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / 8)
    return sorted(best)[1]
for tile in (0, 1):
    _engine.set_option(_engine.OPT_TILE_MODE, tile)
    for rpc in ((0, 8, 12, 16, 24, 32, 64) if tile == 0 else (0,)):
        _engine.set_option(_engine.OPT_ROWS_PER_CHUNK, rpc)
        print(f"tile_mode={tile} rows_per_chunk={rpc}: wavedec2 {t(ptwt_amd.wavedec2, xs, dict(level=4)):.4f} ms   waverec2 {t(ptwt_amd.waverec2, cs, {}):.4f} ms", flush=True)
_engine.set_option(_engine.OPT_ROWS_PER_CHUNK, 0); _engine.set_option(_engine.OPT_TILE_MODE, 0)
