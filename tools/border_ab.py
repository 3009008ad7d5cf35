"""Backward of config 2's wavedec2 (64 x 1024^2 db4 level 3): the border part of the adjoint, one thread per border line (default) against
one thread per border sample (MIFWT_OPT_DEBUG 4096); level by level and the whole forward + backward."""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
E = _engine.ENGINE
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    r = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); r.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(r)[2]
for wav, mode in (('db4', 'reflect'), ('db4', 'periodic'), ('db2', 'reflect'), ('db3', 'reflect'), ('db8', 'reflect')):
    lo, hi = ptwt_amd._wavelets.host_taps(wav)[:2]
    for n in (1024, 515, 261):
        m = (n + len(lo) - 1) // 2
        g = torch.randn(64, 4, m, m, device='cuda')
        f = lambda: E.analysis_adjoint(g, (n, n), lo, hi, _engine.MODE_IDS[mode])
        z = lambda: E.analysis_adjoint(g, (n, n), lo, hi, _engine.MODE_IDS['zero'])
        out = []
        for dbg, ex in ((0, 0), (4096, 0)):
            _engine.set_option(_engine.OPT_DEBUG, dbg); _engine.set_option(15, ex); out.append(t(f))
        _engine.set_option(_engine.OPT_DEBUG, 0); _engine.set_option(15, 0)
        print(f'{wav} {mode} adjoint of a level on 64 x {n}^2: default {out[0]:.1f} us, one thread per sample {out[1]:.1f} us, zero mode (no border) {t(z):.1f} us', flush=True)
x = torch.randn(64, 1024, 1024, device='cuda', requires_grad=True)
def fb():
    c = ptwt_amd.wavedec2(x, 'db4', mode='reflect', level=3)
    s = c[0].sum() + sum(d.sum() for lev in c[1:] for d in lev)
    s.backward(); x.grad = None
for dbg in (0, 4096):
    _engine.set_option(_engine.OPT_DEBUG, dbg)
    print(f'wavedec2 db4 level 3 forward + sum + backward, debug {dbg}: {t(fb, 10):.1f} us')
_engine.set_option(_engine.OPT_DEBUG, 0)
