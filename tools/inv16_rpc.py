"""Finest synthesis level of config 4 (64 x 4096^2 db8, 16 taps) on the streaming kernel: output rows per task."""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
E = _engine.ENGINE
def t(fn, n=6):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    r = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); r.append(e0.elapsed_time(e1) / n)
    return sorted(r)[2]
bank = ptwt_amd._wavelets.host_taps('db8')
rlo, rhi = bank[2], bank[3]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
M = 2055
a = torch.randn(B, M, M, device='cuda')
det = [torch.randn(B, M, M, device='cuda') for _ in range(3)]
def f():
    return E.synthesis(a, det, rlo, rhi, (4096, 4096))
y0 = f().clone()
byt = (4 * B * M * M + B * 4096 * 4096) * 4
for rep in range(2):
    for rpc in (0, 16, 32, 48, 64, 96, 128, 256, 512, 1024, 2048, 4096):
        _engine.set_option(_engine.OPT_ROWS_PER_CHUNK, rpc)
        ms = t(f)
        same = torch.equal(f(), y0)
        print(f'rows per task {rpc:5d}: {ms:.3f} ms  {byt / ms / 1e9:.2f} TB/s = {byt / ms / 8e9:.3f}  bit-identical {same}', flush=True)
_engine.set_option(_engine.OPT_ROWS_PER_CHUNK, 0)
