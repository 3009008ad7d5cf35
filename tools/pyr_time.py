"""Time wavedec2 db4 L3 on 64x1024^2 (config 2) back to back: whole call and the pyramid kernel alone."""
import sys, time, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
wav = sys.argv[1] if len(sys.argv) > 1 else 'db4'
lev = int(sys.argv[2]) if len(sys.argv) > 2 else 3
shape = tuple(int(v) for v in sys.argv[3].split('x')) if len(sys.argv) > 3 else (64, 1024, 1024)
seg = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dbg = int(sys.argv[5]) if len(sys.argv) > 5 else 0
if dbg: _engine.set_option(11, dbg)
pm = int(sys.argv[6]) if len(sys.argv) > 6 else 0
nb = int(sys.argv[7]) if len(sys.argv) > 7 else 0
if nb: _engine.set_option(2, nb)
if pm: _engine.set_option(12, pm)
if seg: _engine.set_option(_engine.OPT_PAIR_ROWS, seg)
xs = [torch.randn(*shape, device='cuda') for _ in range(3)]
for i in range(30): ptwt_amd.wavedec2(xs[i % 3], wav, level=lev)
torch.cuda.synchronize()
res = []
for rnd in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(100): ptwt_amd.wavedec2(xs[i % 3], wav, level=lev)
    e1.record(); torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / 100)
res.sort()
L = len(ptwt_amd._fwt.host_taps(wav)[0])
n = list(shape[1:]); out = 0
for l in range(lev):
    n = [(v + L - 1) // 2 for v in n]; out += 3 * n[0] * n[1]
out += n[0] * n[1]
byts = 4 * shape[0] * (shape[1] * shape[2] + out)
print(f"{wav} L{lev} {shape} seg={seg} dbg={dbg} pyramid_mode={pm} nbuf={nb}: median {res[3]*1e3:.1f} us  min {res[0]*1e3:.1f} us  -> {byts/res[3]/1e6:.0f} GB/s compulsory = {byts/res[3]/8e9:.3f} of 8 TB/s")
