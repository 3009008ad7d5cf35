"""Segment balance and level-2 schedule of kernel 16 on config 2: whole calls with explicit (first, inner) segment rows
(MIFWT_OPT_PYR_SEG0_ROWS / _SEG_ROWS) and with the level-2 waves' old schedule (MIFWT_OPT_DEBUG 8192), same run."""
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
wav = sys.argv[1] if len(sys.argv) > 1 else 'db4'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
xs = [torch.randn(B, 1024, 1024, device='cuda') for _ in range(3)]
i = [0]
def f():
    i[0] += 1; return ptwt_amd.wavedec2(xs[i[0] % 3], wav, level=3)
held = [None, None, None]
def frot():
    i[0] += 1; held[i[0] % 3] = ptwt_amd.wavedec2(xs[i[0] % 3], wav, level=3)
ref = [c.clone() if torch.is_tensor(c) else [v.clone() for v in c] for c in f()]
def same():
    c = f(); torch.cuda.synchronize()
    ok = torch.equal(c[0], ref[0])
    for a, b in zip(c[1:], ref[1:]): ok = ok and all(torch.equal(u, v) for u, v in zip(a, b))
    return ok
combos = [(0, 0, 0), (0, 0, 8192), (0, 0, 4096), (0, 0, 4096 + 8192)]
for a0, r in ((34, 32), (35, 32), (33, 32), (34, 33), (36, 32), (35, 31), (36, 31), (34, 31), (33, 33), (35, 33)): combos.append((a0, r, 0))
combos += [(0, 0, 0), (0, 0, 4096 + 8192)]
for a0, r, dbg in combos:
    _engine.set_option(13, a0); _engine.set_option(14, r); _engine.set_option(11, dbg)
    ok = same()
    m, lo = t(f)
    mr, lor = t(frot)
    held[:] = [None, None, None]
    print(f'{wav} B={B} seg0={a0} seg={r} dbg={dbg}: same output {m:.1f} (min {lo:.1f}) us; rotating outputs {mr:.1f} (min {lor:.1f}) us; bit-identical to the first call: {ok}', flush=True)
_engine.set_option(13, 0); _engine.set_option(14, 0); _engine.set_option(11, 0)
