"""Time the chunked multi-level 1-D analysis launch (mifwt_dwt1_fwd_long) alone and the whole wavedec call.
usage: long1d_time.py [wavelet] [level] [BxN] [mode] [opt1]"""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
wav = sys.argv[1] if len(sys.argv) > 1 else 'db5'
lev = int(sys.argv[2]) if len(sys.argv) > 2 else 10
shape = tuple(int(v) for v in sys.argv[3].split('x')) if len(sys.argv) > 3 else (32, 1000000)
mode = sys.argv[4] if len(sys.argv) > 4 else 'periodic'
if len(sys.argv) > 5: _engine.set_option(1, int(sys.argv[5]))
if len(sys.argv) > 6: _engine.set_option(2, int(sys.argv[6]))
if len(sys.argv) > 7: _engine.set_option(11, int(sys.argv[7]))
xs = [torch.randn(*shape, device='cuda') for _ in range(3)]
taps = ptwt_amd._wavelets.host_taps(wav)
mid = _engine.MODE_IDS[mode]
def t(fn, n=100):
    for i in range(10): fn(xs[i % 3])
    torch.cuda.synchronize()
    res = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n): fn(xs[i % 3])
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / n * 1e3)
    res.sort()
    return res[2], res[0]
bufs = _engine.ENGINE.analysis_tail(xs[0], taps[0], taps[1], mid, lev)
k = len(bufs)
L = len(taps[0])
n, out = shape[1], 0
for l in range(k):
    n = (n + L - 1) // 2; out += n
byts = 4 * shape[0] * (shape[1] + out + n)
med, mn = t(lambda x: _engine.ENGINE.analysis_tail(x, taps[0], taps[1], mid, lev))
print(f"{wav} {shape} {mode}: long kernel ({k} levels) median {med:.1f} us min {mn:.1f} us -> {byts/med/1e3:.0f} GB/s = {byts/med/8e6:.3f} of 8 TB/s")
med, mn = t(lambda x: ptwt_amd.wavedec(x, wav, mode=mode, level=lev))
print(f"   whole wavedec level {lev}: median {med:.1f} us min {mn:.1f} us")
