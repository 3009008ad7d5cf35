"""XCD-parity budgets of kernel 16's schedule (MIFWT_OPT_EXP bits 20-23: delta = 0.005 (v - 1); 0 = the default 0.02): whole calls."""
import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import ptwt_amd
from ptwt_amd import _engine
from test_pyr_schedule import _schedule
def t(fn, n=60):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    r = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); r.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(r)[3], min(r)
xs = [torch.randn(64, 1024, 1024, device='cuda') for _ in range(3)]
i = [0]
def f():
    i[0] += 1; return ptwt_amd.wavedec2(xs[i[0] % 3], 'db4', level=3)
held = [None, None, None]
def frot():
    i[0] += 1; held[i[0] % 3] = ptwt_amd.wavedec2(xs[i[0] % 3], 'db4', level=3)
for rep in range(2):
    for v in (1, 3, 5, 7, 9, 11):
        _engine.set_option(_engine.OPT_EXP, v << 20)
        cuts, hn = _schedule(64, 1024, 1024, 8, 3)
        rows = tuple(int(x) for x in (cuts[1:5] - cuts[0:4]))
        m, lo = t(f); mr, lor = t(frot); held[:] = [None] * 3
        print(f'delta {0.005 * (v - 1):.3f}: rows {rows}: same output {m:.1f} (min {lo:.1f}) us; rotating {mr:.1f} (min {lor:.1f}) us', flush=True)
_engine.set_option(_engine.OPT_EXP, 0)
