"""Non-temporal sub-band / output stores (MIFWT_OPT_NT_STORE) on the big working sets: config 3 (3-D bricks) and config 4 (streaming 2-D)."""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    res = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / n)
    return min(res)
x3 = [torch.randn(8, 256, 256, 256, device='cuda') for _ in range(2)]
c3 = [ptwt_amd.wavedec3(x, 'db2', level=3, mode='zero') for x in x3]
i = [0]
def nxt():
    i[0] += 1
    return i[0] % 2
for rnd in range(2):
    for nt in (0, 1):
        _engine.set_option(_engine.OPT_NT_STORE, nt)
        a = t(lambda: ptwt_amd.wavedec3(x3[nxt()], 'db2', level=3, mode='zero'))
        b = t(lambda: ptwt_amd.waverec3(c3[nxt()], 'db2'))
        print(f"round {rnd} nt {nt}: config 3 wavedec3 {a:.4f} ms  waverec3 {b:.4f} ms", flush=True)
del x3, c3; torch.cuda.empty_cache()
x4 = [torch.randn(64, 4096, 4096, device='cuda') for _ in range(2)]
for rnd in range(2):
    for nt in (0, 1):
        _engine.set_option(_engine.OPT_NT_STORE, nt)
        a = t(lambda: ptwt_amd.wavedec2(x4[nxt()], 'db8', level=4), n=4)
        a1 = t(lambda: ptwt_amd.wavedec2(x4[nxt()], 'db8', level=1), n=4)
        print(f"round {rnd} nt {nt}: config 4 slice wavedec2 level 4 {a:.4f} ms, level 1 alone {a1:.4f} ms", flush=True)
_engine.set_option(_engine.OPT_NT_STORE, 0)
