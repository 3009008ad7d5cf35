"""Cache policies of the matrix-core walk kernels (MIFWT_OPT_DEBUG 8 = write-through stores, 16 = non-temporal stores, 32 = non-temporal
chunk requests): level 1 of the config-5 slice, analysis and synthesis, alternating rounds."""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
ptwt_amd.set_half_storage(True)
x = torch.randn(32, 8192, 8192, device='cuda').half()
cs = ptwt_amd.wavedec2(x, 'sym16', mode='reflect', level=1)
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    res = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / n)
    return min(res)
for rnd in range(2):
    for dbg in (0, 8, 16, 64):
        _engine.set_option(_engine.OPT_DEBUG, dbg)
        ta = t(lambda: ptwt_amd.wavedec2(x, 'sym16', mode='reflect', level=1))
        ts = t(lambda: ptwt_amd.waverec2(cs, 'sym16'))
        print(f"round {rnd} debug {dbg:2d}: analysis {ta:.3f} ms   synthesis {ts:.3f} ms", flush=True)
_engine.set_option(_engine.OPT_DEBUG, 0)
