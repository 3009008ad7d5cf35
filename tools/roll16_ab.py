"""Config 4 slice (wavedec2 db8 L4 on 64x4096^2): two-level rolling-strip launches (10 .. 16 taps, round 4) against per-level launches;
whole calls back to back + per-launch events.  argv: [wavelet] [level] [BxHxW]"""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
wav = sys.argv[1] if len(sys.argv) > 1 else 'db8'
lev = int(sys.argv[2]) if len(sys.argv) > 2 else 4
shape = tuple(int(v) for v in sys.argv[3].split('x')) if len(sys.argv) > 3 else (64, 4096, 4096)
xs = [torch.randn(*shape, device='cuda') for _ in range(3)]
for pm, name in ((2, 'per level'), (0, 'auto'), (3, 'rolling strips wherever they apply')):
    _engine.set_option(_engine.OPT_PAIR_MODE, pm)
    for i in range(6): ptwt_amd.wavedec2(xs[i % 3], wav, level=lev)
    torch.cuda.synchronize()
    res = []
    for rnd in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(12): ptwt_amd.wavedec2(xs[i % 3], wav, level=lev)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 12)
    res.sort()
    _engine.level_events = []
    for i in range(6): ptwt_amd.wavedec2(xs[i % 3], wav, level=lev)
    torch.cuda.synchronize()
    ev, _engine.level_events = _engine.level_events, None
    per = {}
    for tag, kid, ext, s, e in ev: per.setdefault((kid, tuple(ext)), []).append(s.elapsed_time(e))
    print(f"{wav} L{lev} {shape} pair_mode={pm} ({name}): median {res[2]:.4f} ms  min {res[0]:.4f};  launches: " +
          ", ".join(f"id{k} {'x'.join(map(str, x))}: {sum(v)/len(v):.4f}" for (k, x), v in per.items()), flush=True)
_engine.set_option(_engine.OPT_PAIR_MODE, 0)
