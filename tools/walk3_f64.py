"""f64 volumes: the walk kernel (id 24) against the composed route (id 5), level by level on config 3's shapes."""
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
wav = sys.argv[1] if len(sys.argv) > 1 else 'db2'
lo, hi = ptwt_amd._wavelets.host_taps(wav)[:2]
for n in (256, 129, 66):
    x = torch.randn(8, n, n, n, device='cuda', dtype=torch.float64)
    f = lambda: E.analysis(x, lo, hi, _engine.MODE_IDS['zero'])
    m = (n + len(lo) - 1) // 2
    byt = 8 * 8 * (n ** 3 + 8 * m ** 3)
    for tm, rows, pf in ((2, 0, 0), (0, 0, 0), (4, 4, 0), (4, 2, 0)):
        _engine.set_option(_engine.OPT_TILE_MODE, tm); _engine.set_option(_engine.OPT_TILE_ROWS, rows); _engine.set_option(_engine.OPT_PREFETCH_PAIRS, pf)
        _engine.level_events = []
        f(); kid = _engine.level_events[0][1]; _engine.level_events = None
        us = t(f)
        print(f'{wav} 8 x {n}^3 f64: tile mode {tm} rows {rows} slices ahead {pf or "default"}: kernel id {kid}, {us:.1f} us = {byt / us / 8e6:.3f} of 8 TB/s', flush=True)
    del x
for k in (_engine.OPT_TILE_MODE, _engine.OPT_TILE_ROWS, _engine.OPT_PREFETCH_PAIRS): _engine.set_option(k, 0)
rlo, rhi = ptwt_amd._wavelets.host_taps(wav)[2:4]
for n in (256, 129, 66):
    m = (n + len(lo) - 1) // 2
    a = torch.randn(8, m, m, m, device='cuda', dtype=torch.float64)
    det = [torch.randn(8, m, m, m, device='cuda', dtype=torch.float64) for _ in range(7)]
    f = lambda: E.synthesis(a, det, rlo, rhi, (n, n, n))
    byt = 8 * 8 * (n ** 3 + 8 * m ** 3)
    for tm, pf in ((2, 0), (0, 0), (0, 1), (0, 2), (0, 3)):
        _engine.set_option(_engine.OPT_TILE_MODE, tm); _engine.set_option(_engine.OPT_PREFETCH_PAIRS, pf)
        _engine.level_events = []
        f(); kid = _engine.level_events[0][1]; _engine.level_events = None
        us = t(f)
        print(f'{wav} synthesis -> 8 x {n}^3 f64: tile mode {tm} slices ahead {pf or "default"}: kernel id {kid}, {us:.1f} us = {byt / us / 8e6:.3f} of 8 TB/s', flush=True)
    del a, det
for k in (_engine.OPT_TILE_MODE, _engine.OPT_TILE_ROWS, _engine.OPT_PREFETCH_PAIRS): _engine.set_option(k, 0)
x = torch.randn(8, 256, 256, 256, device='cuda', dtype=torch.float64)
g = lambda: ptwt_amd.wavedec3(x, wav, mode='zero', level=3)
print(f'wavedec3 {wav} level 3 on 8 x 256^3 f64, auto: {t(g):.1f} us')
c = g()
h = lambda: ptwt_amd.waverec3(c, wav)
print(f'waverec3 {wav} level 3 on 8 x 256^3 f64, auto: {t(h):.1f} us')
