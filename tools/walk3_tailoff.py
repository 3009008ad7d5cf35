"""Level 1 of config 3 on the 3-D walking analysis kernel: loads / stores switched off (MIFWT_OPT_DEBUG 1, 2), then 3-8 column strips
(MIFWT_OPT_EXP)."""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
E = _engine.ENGINE
def t(fn, n=40):
    for _ in range(8): fn()
    torch.cuda.synchronize()
    r = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); r.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(r)[3]
lo, hi = ptwt_amd._wavelets.host_taps('db2')[:2]
xs = [torch.randn(8, 256, 256, 256, device='cuda') for _ in range(3)]
i = [0]
def f():
    i[0] += 1; return E.analysis(xs[i[0] % 3], lo, hi, _engine.MODE_IDS['zero'])
for rep in range(2):
    for dbg in (0, 1, 2, 3):
        _engine.set_option(11, dbg)
        print(f'level 1 of 8 x 256^3 db2, debug {dbg} (1: no stores, 2: no loads): {t(f):.1f} us', flush=True)
_engine.set_option(11, 0)
for ns in (0, 3, 4, 5, 6, 8):
    _engine.set_option(15, ns)
    print(f'level 1 of 8 x 256^3 db2, {ns} strips: {t(f):.1f} us', flush=True)
_engine.set_option(15, 0)
