"""What bounds the walking matrix-core kernel?  Level 1 of the config-5 slice with stores / loads / matrix work switched off (MIFWT_OPT_DEBUG 1 / 2 / 4)."""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
ptwt_amd.set_half_storage(True)
x = torch.randn(32, 8192, 8192, device='cuda').half()
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
for dbg in [int(v) for v in sys.argv[1:]] or (0, 1, 2, 4, 3, 5, 6, 7, 0):  # (8 / 16 = store policies, see mfma_policy_ab.py)
    _engine.set_option(_engine.OPT_DEBUG, dbg)
    print(f"debug {dbg} ({'no stores ' if dbg & 1 else ''}{'no loads ' if dbg & 2 else ''}{'no matrix work' if dbg & 4 else ''}): {t(lambda: ptwt_amd.wavedec2(x, 'sym16', mode='reflect', level=1)):.3f} ms", flush=True)
_engine.set_option(_engine.OPT_DEBUG, 0)
