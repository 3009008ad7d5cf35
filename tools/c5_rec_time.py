"""fswaverec2 / waverec2 of the config-5 slice (32 x 8192^2 f16 sym16 level 5): time per call and per level kernel."""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
ptwt_amd.set_half_storage(True)
x = torch.randn(32, 8192, 8192, device='cuda').half()
cs = ptwt_amd.fswavedec2(x, 'sym16', mode='reflect', level=5)
del x
def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    res = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / n)
    return min(res)
import os
_engine.set_option(7, int(os.environ.get('MFMA_MODE', '0')))
_engine.level_events = []
y = ptwt_amd.fswaverec2(cs, 'sym16'); torch.cuda.synchronize()
ev = _engine.level_events; _engine.level_events = None
print('launches (tag, kernel id, ms):', [(e[0], e[1], e[2], round(e[3].elapsed_time(e[4]), 3)) for e in ev])
del y
print(f"fswaverec2: {t(lambda: ptwt_amd.fswaverec2(cs, 'sym16')):.3f} ms per call")
