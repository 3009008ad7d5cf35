"""3-D synthesis bricks: row-brick index skewed by the depth-brick index (MIFWT_OPT_DEBUG bits 4-7 = the skew) — outputs with power-of-two
slice pitches put the bricks of neighbouring depths on the same memory channels."""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    res = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / n)
    return min(res)
x = torch.randn(8, 256, 256, 256, device='cuda')
c1 = ptwt_amd.wavedec3(x, 'db2', level=1, mode='zero')
c3 = ptwt_amd.wavedec3(x, 'db2', level=3, mode='zero')
ref = ptwt_amd.waverec3(c3, 'db2')
for rnd in range(2):
    for skew in (0, 1, 3, 5, 7, 11):
        _engine.set_option(_engine.OPT_DEBUG, skew << 4)
        y = ptwt_amd.waverec3(c3, 'db2')
        same = torch.equal(y, ref)
        print(f"round {rnd} skew {skew:2d}: level 1 {t(lambda: ptwt_amd.waverec3(c1, 'db2')):.4f} ms   three levels {t(lambda: ptwt_amd.waverec3(c3, 'db2')):.4f} ms   {'same result' if same else 'DIFFERENT'}", flush=True)
_engine.set_option(_engine.OPT_DEBUG, 0)
