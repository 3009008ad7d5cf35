"""The walking matrix-core analysis kernel on the planes of config 5's levels (32 x 8192^2 / 4111^2 / 2071^2 / 1051^2 f16, sym16): whole level
and with stores / loads / matrix work switched off (MIFWT_OPT_DEBUG 1 / 2 / 4), dense rows against a 128-byte aligned view as input."""
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
    m = (n + 31) // 2
    byt = 32 * (n * n + 4 * m * m) * 2
    pitch = -(-n * 2 // 128) * 64
    for name, x in (('dense', torch.randn(32, n, n, device='cuda').half()), ('128-byte rows', torch.randn(32, n, pitch, device='cuda').half()[..., :n])):
        out = []
        for dbg in (0, 1, 2, 4, 7):
            _engine.set_option(_engine.OPT_DEBUG, dbg)
            out.append(t(lambda: ptwt_amd.wavedec2(x, 'sym16', mode='reflect', level=1)))
        _engine.set_option(_engine.OPT_DEBUG, 0)
        print(f'32 x {n}^2 ({name}): {out[0]:.3f} ms = {byt / out[0] / 8e9:.3f} of 8 TB/s; no stores {out[1]:.3f}, no loads {out[2]:.3f}, no matrix work {out[3]:.3f}, none of them {out[4]:.3f}', flush=True)
        del x
