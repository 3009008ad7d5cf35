"""Where does the streaming analysis launch (kernel id 16) pay?  wavedec2 per call with MIFWT_OPT_PYRAMID_MODE 0 (auto) / 1 (wherever it can
run) / 2 (multi-level launches off)."""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    res = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(res)[2]
for shape, wav, lev in [((256, 256, 256), 'db4', 3), ((256, 384, 384), 'db4', 3), ((128, 448, 448), 'db2', 3), ((64, 640, 480), 'db3', 3), ((128, 512, 512), 'db4', 4),
                        ((16, 1024, 1024), 'db4', 3), ((4, 1024, 1024), 'db4', 3), ((16, 1280, 720), 'sym4', 4), ((8, 1500, 1500), 'db2', 3), ((16, 2048, 2048), 'sym4', 4), ((64, 1000, 1000), 'db4', 3)]:
    xs = [torch.randn(*shape, device='cuda') for _ in range(3)]
    i = [0]
    def fwd():
        i[0] += 1
        return ptwt_amd.wavedec2(xs[i[0] % 3], wav, level=lev)
    out = []
    for mode in (0, 1, 2):
        _engine.set_option(_engine.OPT_PYRAMID_MODE, mode)
        _engine.level_events = []
        c = fwd(); torch.cuda.synchronize()
        kids = [e[1] for e in _engine.level_events]; _engine.level_events = None
        out.append(f"mode {mode}: {t(fwd):7.1f} us {kids}")
    _engine.set_option(_engine.OPT_PYRAMID_MODE, 0)
    byts = 4 * (xs[0].numel() + c[0].numel() + sum(v.numel() for d in c[1:] for v in d))
    print(f"{shape} {wav} L{lev}: " + "   ".join(out) + f"   ({byts/1e6:.0f} MB)")
    del xs, c; torch.cuda.empty_cache()
