"""Where does the streaming synthesis launch (kernel id 22) pay?  waverec2 per call with MIFWT_OPT_PYRAMID_MODE 0 (auto) / 1 (wherever it
can run) / 2 (multi-level launches off), and the kernels that ran."""
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
for shape, wav, lev in [((1024, 128, 128), 'db4', 3), ((256, 256, 256), 'db4', 3), ((512, 192, 192), 'db2', 3), ((256, 384, 384), 'db4', 3), ((128, 512, 512), 'db4', 4), ((256, 512, 512), 'db4', 4),
                        ((64, 640, 480), 'db3', 3), ((16, 1024, 1024), 'db4', 3), ((4, 1024, 1024), 'db4', 3), ((16, 1280, 720), 'sym4', 4), ((64, 1024, 1024), 'haar', 5), ((8, 1500, 1500), 'db2', 3)]:
    cs = [ptwt_amd.wavedec2(torch.randn(*shape, device='cuda'), wav, level=lev) for _ in range(3)]
    i = [0]
    def inv():
        i[0] += 1
        return ptwt_amd.waverec2(cs[i[0] % 3], wav)
    out = []
    for mode in (0, 1, 2):
        _engine.set_option(_engine.OPT_PYRAMID_MODE, mode)
        _engine.level_events = []
        inv(); torch.cuda.synchronize()
        kids = [e[1] for e in _engine.level_events]; _engine.level_events = None
        out.append(f"mode {mode}: {t(inv):7.1f} us {kids}")
    _engine.set_option(_engine.OPT_PYRAMID_MODE, 0)
    byts = 4 * (shape[0] * shape[1] * shape[2] + cs[0][0].numel() + sum(v.numel() for d in cs[0][1:] for v in d))
    print(f"{shape} {wav} L{lev}: " + "   ".join(out) + f"   ({byts/1e6:.0f} MB)")
    del cs; torch.cuda.empty_cache()
