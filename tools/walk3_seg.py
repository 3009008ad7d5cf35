"""wavedec3 / waverec3 config 3 against the depth-segment length of the walking kernels (MIFWT_OPT_ROWS_PER_CHUNK = output slices per segment)."""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
def t(fn, n=40):
    for _ in range(8): fn()
    torch.cuda.synchronize()
    r = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); r.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(r)[3], min(r)
shape = tuple(int(v) for v in sys.argv[1].split('x')) if len(sys.argv) > 1 else (8, 256, 256, 256)
wav = sys.argv[2] if len(sys.argv) > 2 else 'db2'
xs = [torch.randn(*shape, device='cuda') for _ in range(3)]
cs = [ptwt_amd.wavedec3(x, wav, level=3) for x in xs]
i = [0]
def f():
    i[0] += 1; return ptwt_amd.wavedec3(xs[i[0] % 3], wav, level=3)
def g():
    i[0] += 1; return ptwt_amd.waverec3(cs[i[0] % 3], wav)
for rep in range(2):
    for seg in (0, 8, 11, 13, 15, 17, 19, 22, 26, 33):
        _engine.set_option(1, seg)
        a, amin = t(f); b, bmin = t(g)
        print(f'{shape} {wav} slices per segment {seg or "default"}: wavedec3 {a:.1f} (min {amin:.1f}) us, waverec3 {b:.1f} (min {bmin:.1f}) us', flush=True)
_engine.set_option(1, 0)
