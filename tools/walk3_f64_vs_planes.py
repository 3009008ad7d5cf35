"""f64 volumes, analysis level by level: walk kernel (auto) against the composed planes + depth pass route (MIFWT_OPT_TILE_MODE 1)."""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
E = _engine.ENGINE
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    r = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); r.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(r)[2]
for wav in ('db2', 'db3', 'db4', 'db5'):
    bank = ptwt_amd._wavelets.host_taps(wav)
    for shape in ((8, 256, 256, 256), (8, 129, 129, 129), (8, 66, 66, 66), (32, 100, 100, 100), (64, 32, 32, 32), (8, 40, 40, 40)):
        x = torch.randn(*shape, device='cuda', dtype=torch.float64)
        f = lambda: E.analysis(x, bank[0], bank[1], _engine.MODE_IDS['zero'])
        coef = [(n + len(bank[0]) - 1) // 2 for n in shape[1:]]
        a = torch.randn(shape[0], *coef, device='cuda', dtype=torch.float64)
        det = [torch.randn(shape[0], *coef, device='cuda', dtype=torch.float64) for _ in range(7)]
        g = lambda: E.synthesis(a, det, bank[2], bank[3], shape[1:])
        out = []
        for fn in (f, g):
            for tm in (0, 1):
                _engine.set_option(_engine.OPT_TILE_MODE, tm)
                _engine.level_events = []
                fn(); kid = _engine.level_events[0][1]; _engine.level_events = None
                out.append(f'id {kid}: {t(fn):.1f} us')
        _engine.set_option(_engine.OPT_TILE_MODE, 0)
        print(f'{wav} {shape} f64: analysis auto {out[0]}, tile mode 1 {out[1]}; synthesis auto {out[2]}, tile mode 1 {out[3]}', flush=True)
        del x, a, det
