"""Config 2 (wavedec2 db4 L3 on 64x1024^2) whole calls, back to back: results dropped (the caching allocator hands every call the same
output block) and with the last three results kept alive (three rotating output sets).  Environment: MIFWT_LIB, MIFWT_PYRAMID_ROW_ALIGN;
argv: [debug bits] [wavelet] [level] [BxHxW]."""
import os, sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
dbg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
wav = sys.argv[2] if len(sys.argv) > 2 else 'db4'
lev = int(sys.argv[3]) if len(sys.argv) > 3 else 3
shape = tuple(int(v) for v in sys.argv[4].split('x')) if len(sys.argv) > 4 else (64, 1024, 1024)
if dbg: _engine.set_option(11, dbg)
xs = [torch.randn(*shape, device='cuda') for _ in range(3)]
def timed(keep):
    held = [None] * 3
    for i in range(40):
        if keep: held[i % 3] = ptwt_amd.wavedec2(xs[i % 3], wav, level=lev)
        else: ptwt_amd.wavedec2(xs[i % 3], wav, level=lev)  # (result dropped at once: the next call gets the same block)
    torch.cuda.synchronize()
    res = []
    for rnd in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(100):
            if keep: held[i % 3] = ptwt_amd.wavedec2(xs[i % 3], wav, level=lev)
            else: ptwt_amd.wavedec2(xs[i % 3], wav, level=lev)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 100)
    res.sort()
    return res[len(res) // 2], res[0]
a = timed(False); b = timed(True); a2 = timed(False)
r = ptwt_amd.wavedec2(xs[0], wav, level=lev)
print(f"lib={os.environ.get('MIFWT_LIB','libmifwt.so')} align={_engine.PYRAMID_ROW_ALIGN} dbg={dbg} {wav} L{lev} {shape} strides {r[1][0].stride()}: "
      f"same output {a[0]*1e3:.1f} (min {a[1]*1e3:.1f}) / again {a2[0]*1e3:.1f} us;  rotating outputs {b[0]*1e3:.1f} (min {b[1]*1e3:.1f}) us")
