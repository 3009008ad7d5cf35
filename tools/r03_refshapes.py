"""The reference's own published benchmark shapes (examples/speed_tests/timeitconv_2d.py:38-57, timeitconv_2d_separable.py:43-85,
timeitconv_3d.py:54-64): us per call and fraction of the HBM peak on the compulsory bytes, plus the per-launch split."""
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

def flat(c):
    out = []
    for v in c:
        if isinstance(v, torch.Tensor): out.append(v)
        elif isinstance(v, dict): out.extend(v.values())
        else: out.extend(v)
    return out

CASES = [("wavedec2", "waverec2", (32, 1000, 1000), "db5", 5, "periodic"),
         ("fswavedec2", "fswaverec2", (32, 1000, 1000), "db5", 5, "periodic"),
         ("wavedec3", "waverec3", (32, 100, 100, 100), "db5", 3, "periodic"),
         ("fswavedec3", "fswaverec3", (32, 100, 100, 100), "db5", 3, "periodic"),
         ("wavedec2", "waverec2", (32, 1000, 1000), "db5", 5, "reflect"),
         ("wavedec2", "waverec2", (64, 4096, 4096), "db8", 4, "reflect"),
         ("wavedec3", "waverec3", (8, 256, 256, 256), "db2", 3, "zero"),
         ("wavedec2", "waverec2", (64, 1024, 1024), "db4", 3, "reflect")]
for fa, fs, shape, wav, lev, mode in CASES:
    xs = [torch.randn(*shape, device='cuda') for _ in range(3)]
    A, S = getattr(ptwt_amd, fa), getattr(ptwt_amd, fs)
    i = [0]
    def fwd():
        i[0] += 1
        return A(xs[i[0] % 3], wav, level=lev, mode=mode)
    cs = [A(x, wav, level=lev, mode=mode) for x in xs]
    def inv():
        i[0] += 1
        return S(cs[i[0] % 3], wav)
    byts = 4 * (xs[0].numel() + sum(v.numel() for v in flat(cs[0])))
    a, b = t(fwd), t(inv)
    print(f"{fa} {shape} {wav} L{lev} {mode}: analysis {a:8.1f} us ({byts/a/8e6:.3f})   synthesis {b:8.1f} us ({byts/b/8e6:.3f})")
    for name, f in (("fwd", fwd), ("inv", inv)):
        _engine.level_events = []
        for _ in range(10): f()
        torch.cuda.synchronize()
        ev, _engine.level_events = _engine.level_events, None
        agg = {}
        for tag, kid, ext, s, e in ev:
            agg.setdefault((tag, kid, tuple(ext)), []).append(s.elapsed_time(e) * 1e3)
        print("   " + name + ": " + "; ".join('id %d %s %.1f us' % (k[1], 'x'.join(map(str, k[2])), sorted(v)[len(v) // 2]) for k, v in agg.items()))
    del xs, cs
    torch.cuda.empty_cache()
