"""A/B timing of 3-D synthesis on BASELINE config 3 (8 x 256^3 db2, level 1 and 3 levels): the fused brick kernel (id 10) against the
composed route (tile mode 2: fused 2-D planes + depth pass through scratch)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptwt_amd
from ptwt_amd import _engine
dev = torch.device("cuda:0")
wav = os.environ.get("MIFWT_WAVELET", "db2")
xs = [torch.randn(8, 256, 256, 256, device=dev) for _ in range(3)]
for lvl in (1, 3):
    cs = [ptwt_amd.wavedec3(x, wav, level=lvl, mode="zero") for x in xs]
    nbytes = 4 * (xs[0].numel() + cs[0][0].numel() + sum(v.numel() for c in cs[0][1:] for v in c.values()))
    for name, opt5, opt6 in (("fused bricks", 0, 0), ("composed", 2, 0), ("fused bricks", 0, 0), ("bricks 4x16", 0, 8)):
        _engine.set_option(5, opt5)
        _engine.set_option(6, opt6)
        for i in range(3): ptwt_amd.waverec3(cs[i], wav)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(12): ptwt_amd.waverec3(cs[i % 3], wav)
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 12
        print(f"waverec3 {wav} level {lvl} {name:13s} {ms:.4f} ms  ({nbytes / ms / 8e9:.3f} of the HBM peak on the compulsory bytes)")
    _engine.set_option(5, 0)
    _engine.set_option(6, 0)
    del cs
