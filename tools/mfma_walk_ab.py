"""The walking matrix-core analysis kernel (MIFWT_OPT_MFMA_MODE 0) against the tile-at-a-time one (mode 3): bit equality and time per level."""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
ptwt_amd.set_half_storage(True)
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
segs = [int(v) for v in sys.argv[1:]] or [0]
for shape, wav, mode in [((32, 8192, 8192), 'sym16', 'reflect'), ((32, 4111, 4111), 'sym16', 'reflect'), ((32, 2071, 2071), 'sym16', 'reflect'),
                         ((32, 1051, 1051), 'sym16', 'reflect'), ((32, 541, 541), 'sym16', 'symmetric'), ((8, 3000, 2000), 'db10', 'zero')]:
    x = torch.randn(*shape, device='cuda').half()
    out = {}
    line = []
    for m in (3, 4, 0):
        _engine.set_option(7, m)
        for seg in (segs if m == 4 else [0]):
            _engine.set_option(6, seg)
            c = ptwt_amd.wavedec2(x, wav, mode=mode, level=1)
            torch.cuda.synchronize()
            ms = t(lambda: ptwt_amd.wavedec2(x, wav, mode=mode, level=1))
            line.append(f"mode {m} seg {seg}: {ms:.3f} ms")
            if m == 3: ref = c
            else:
                same = torch.equal(c[0], ref[0]) and all(torch.equal(a, b) for a, b in zip(c[1], ref[1]))
                line.append("bit-identical" if same else f"DIFFERENT (max {max(float((a.float()-b.float()).abs().max()) for a, b in zip([c[0], *c[1]], [ref[0], *ref[1]])):.3e})")
    _engine.set_option(6, 0); _engine.set_option(7, 0)
    print(shape, wav, mode, '  '.join(line), flush=True)
    del x, c, ref; torch.cuda.empty_cache()
