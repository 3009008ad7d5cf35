"""wavedec2 / waverec2 over small-plane batches: the one-launch kernels (ids 20 / 21) against the per-level / pair kernels."""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
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
SHAPES = [((4096, 64, 64), 'db2', 3), ((1024, 128, 128), 'db4', 3), ((16384, 32, 32), 'db2', 2), ((2048, 96, 96), 'sym4', 3), ((8192, 48, 48), 'haar', 4), ((512, 128, 128), 'db2', 5), ((32768, 16, 16), 'haar', 2), ((8192, 40, 40), 'db4', 3), ((4096, 64, 64), 'db4', 3), ((4096, 72, 72), 'db2', 3), ((2048, 88, 88), 'db4', 3)]
if len(sys.argv) > 1 and sys.argv[1] == 'probe':  # the one-workgroup-per-CU boundary
    SHAPES = [((2048, 80, 80), 'sym4', 3), ((2048, 88, 88), 'db4', 3), ((2048, 96, 96), 'db2', 3), ((1024, 112, 112), 'db4', 3), ((1024, 120, 120), 'db2', 3)]
    _engine.set_option(_engine.OPT_PYRAMID_MODE, 3)
if len(sys.argv) > 6 and sys.argv[1] == 'one':  # one b h w wavelet level [...]
    a_ = sys.argv[2:]
    SHAPES = [((int(a_[i]), int(a_[i + 1]), int(a_[i + 2])), a_[i + 3], int(a_[i + 4])) for i in range(0, len(a_) - 4, 5)]
if len(sys.argv) > 1 and sys.argv[1] == 'one':
    _engine.set_option(_engine.OPT_PYRAMID_MODE, 3)
MODE = _engine.get_option(_engine.OPT_PYRAMID_MODE) if hasattr(_engine, 'get_option') else (3 if len(sys.argv) > 1 and sys.argv[1] in ('probe', 'one') else 0)
for shape, wav, lev in SHAPES:
    xs = [torch.randn(*shape, device='cuda') for _ in range(3)]
    cs = [ptwt_amd.wavedec2(x, wav, level=lev) for x in xs]
    i = [0]
    def inv():
        i[0] += 1
        return ptwt_amd.waverec2(cs[i[0] % 3], wav)
    ncoef = cs[0][0].numel() + sum(t_.numel() for d in cs[0][1:] for t_ in d)
    byts = 4 * (xs[0].numel() + ncoef)
    def fwd():
        i[0] += 1
        return ptwt_amd.wavedec2(xs[i[0] % 3], wav, level=lev)
    a, f = t(inv), t(fwd)
    _engine.set_option(_engine.OPT_PYRAMID_MODE, 2)
    b, f2 = t(inv), t(fwd)
    _engine.set_option(_engine.OPT_PYRAMID_MODE, MODE)
    print(f"{shape} {wav} L{lev}: wavedec2 {f:6.1f} us ({byts/f/8e6:.3f} of the HBM peak; level by level / pairs {f2:6.1f} us)   "
          f"waverec2 {a:6.1f} us ({byts/a/8e6:.3f}; level by level / pairs {b:6.1f} us)")
