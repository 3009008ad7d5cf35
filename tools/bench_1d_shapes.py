"""wavedec / waverec over typical 1-D batch shapes: ms per call and fraction of the HBM peak on the compulsory bytes."""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
if len(sys.argv) > 1: _engine.set_option(11, int(sys.argv[1]))
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    res = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(res)[1]
for shape, wav, lev, dt in [((1024, 16384), 'db4', 8, torch.float32), ((4096, 4096), 'db4', 6, torch.float32), ((128, 65536), 'db4', 10, torch.float32),
                            ((16384, 1024), 'db2', 5, torch.float32), ((256, 262144), 'sym8', 10, torch.float32), ((64, 1000000), 'db5', 10, torch.float64),
                            ((1024, 16384), 'db4', 8, torch.float64)]:
    xs = [torch.randn(*shape, device='cuda', dtype=dt) for _ in range(3)]
    i = [0]
    def fwd():
        i[0] += 1
        return ptwt_amd.wavedec(xs[i[0] % 3], wav, level=lev)
    cs = [ptwt_amd.wavedec(x, wav, level=lev) for x in xs]
    def inv():
        i[0] += 1
        return ptwt_amd.waverec(cs[i[0] % 3], wav)
    byts = xs[0].element_size() * (xs[0].numel() + sum(c.numel() for c in cs[0]))
    a, b = t(fwd), t(inv)
    print(f"{shape} {wav} L{lev} {str(dt)[6:]}: wavedec {a:8.1f} us ({byts/a/8e6:.3f})   waverec {b:8.1f} us ({byts/b/8e6:.3f})")
