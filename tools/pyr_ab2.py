"""A/B of MIFWT_OPT_DEBUG switches of kernel 16 on config 2, alternating, same run (timings only: some switches break the results)."""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
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
wav = 'db4'
xs = [torch.randn(64, 1024, 1024, device='cuda') for _ in range(3)]
i = [0]
def f():
    i[0] += 1; return ptwt_amd.wavedec2(xs[i[0] % 3], wav, level=3)
held = [None, None, None]
def frot():
    i[0] += 1; held[i[0] % 3] = ptwt_amd.wavedec2(xs[i[0] % 3], wav, level=3)
dbgs = [int(v) for v in sys.argv[1].split(',')]
opt = int(sys.argv[2]) if len(sys.argv) > 2 else 11
for rep in range(2):
    for dbg in dbgs:
        _engine.set_option(opt, dbg)
        m, lo = t(f)
        mr, lor = t(frot)
        held[:] = [None, None, None]
        print(f'{wav} option {opt} = {dbg}: same output {m:.1f} (min {lo:.1f}) us; rotating outputs {mr:.1f} (min {lor:.1f}) us', flush=True)
_engine.set_option(opt, 0)
