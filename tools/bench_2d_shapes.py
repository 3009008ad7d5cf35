"""wavedec2 / waverec2 over typical 2-D batch shapes: us per call and fraction of the HBM peak on the compulsory bytes."""
import os, sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
NSHAPES = int(os.environ.get('MIFWT_SHAPES', '99'))
FIRST = int(os.environ.get('MIFWT_FIRST', '0'))
def t(fn, n=20):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    res = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(res)[1]
for shape, wav, lev, dt in [((4096, 64, 64), 'db2', 3, torch.float32), ((1024, 128, 128), 'db4', 3, torch.float32), ((256, 256, 256), 'db4', 3, torch.float32),
                            ((256, 512, 512), 'db4', 4, torch.float32), ((64, 1024, 1024), 'db4', 3, torch.float32), ((64, 1024, 1024), 'haar', 5, torch.float32),
                            ((16, 2048, 2048), 'sym4', 4, torch.float32), ((48, 3, 512, 768), 'db3', 3, torch.float32), ((64, 1000, 1000), 'db4', 3, torch.float32),
                            ((32, 1024, 1024), 'db4', 3, torch.float64), ((64, 1024, 1024), 'db4', 3, 'periodic')][FIRST:NSHAPES]:
    mode = 'reflect'
    if isinstance(dt, str):
        mode, dt = dt, torch.float32
    xs = [torch.randn(*shape, device='cuda', dtype=dt) for _ in range(3)]
    i = [0]
    def fwd():
        i[0] += 1
        return ptwt_amd.wavedec2(xs[i[0] % 3], wav, level=lev, mode=mode)
    cs = [ptwt_amd.wavedec2(x, wav, level=lev, mode=mode) for x in xs]
    def inv():
        i[0] += 1
        return ptwt_amd.waverec2(cs[i[0] % 3], wav)
    ncoef = cs[0][0].numel() + sum(t_.numel() for d in cs[0][1:] for t_ in d)
    byts = xs[0].element_size() * (xs[0].numel() + ncoef)
    a, b = t(fwd), t(inv)
    _engine.set_option(_engine.OPT_PYRAMID_MODE, 2)  # one launch per level (pairs allowed)
    a2, b2 = t(fwd), t(inv)
    _engine.set_option(_engine.OPT_PYRAMID_MODE, 3)  # the small-plane launches wherever they can run
    a3, b3 = t(fwd), t(inv)
    _engine.set_option(_engine.OPT_PYRAMID_MODE, 0)
    print(f"{shape} {wav} L{lev} {str(dt)[6:]} {mode}: wavedec2 {a:8.1f} us ({byts/a/8e6:.3f}; multi-level launches off {a2:8.1f}, small-plane launches forced {a3:8.1f} us)   waverec2 {b:8.1f} us ({byts/b/8e6:.3f}; off {b2:8.1f}, forced {b3:8.1f})")
