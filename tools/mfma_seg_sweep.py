"""The walking matrix-core analysis kernel: tiles per unit (MIFWT_OPT_TILE_ROWS) on the planes of config 5's levels."""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
ptwt_amd.set_half_storage(True)
def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    res = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / n)
    return min(res)
for n in (8192, 4111, 2071, 1051):
    x = torch.randn(32, n, n, device='cuda').half()
    m = (n + 31) // 2
    tiles_r = (m + 15) // 16
    out = []
    for seg in (0, 4, 8, 16, 33, 65, 130, 257):
        if seg > tiles_r and seg != 0 and out and seg // 2 > tiles_r: continue
        _engine.set_option(_engine.OPT_TILE_ROWS, seg)
        out.append(f'{seg or "default"}: {t(lambda: ptwt_amd.wavedec2(x, "sym16", mode="reflect", level=1)):.3f}')
    _engine.set_option(_engine.OPT_TILE_ROWS, 0)
    print(f'32 x {n}^2 ({tiles_r} tile rows), ms by tiles per unit: ' + ', '.join(out), flush=True)
    del x
