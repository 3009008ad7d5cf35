"""A/B timing of the small-plane pyramid kernel with MIFWT_OPT_DEBUG bits (results are wrong with any bit set)."""
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
for shape, wav, lev in [((4096, 64, 64), 'db4', 3), ((8192, 40, 40), 'db4', 3)]:
    xs = [torch.randn(*shape, device='cuda') for _ in range(3)]
    i = [0]
    def fwd():
        i[0] += 1
        return ptwt_amd.wavedec2(xs[i[0] % 3], wav, level=lev)
    for bits, name in [(0, 'full'), (1, 'no stores'), (2, 'no park'), (3, 'no park, no stores'), (64, 'no horizontal interior'),
                       (128, 'no vertical interior'), (32, 'no pad fills'), (64 + 128, 'no filter passes'), (1 + 2 + 32 + 64 + 128, 'barriers and level set-up only'), (1 + 2 + 64 + 128, 'nothing (host floor)')]:
        _engine.set_option(_engine.OPT_DEBUG, bits)
        print(shape, wav, name, "%.1f us" % t(fwd))
    _engine.set_option(_engine.OPT_DEBUG, 0)
