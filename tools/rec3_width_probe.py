"""How much does the third (two-lane) column brick of the 3-D synthesis kernel cost?  One level of waverec3 db2 on 8 volumes of
256 x 256 x W for W = 252 (two bricks of 126 columns exactly), 256 (two bricks + a third with 4 columns), 254."""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
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
for W in (252, 256, 254, 378, 384):
    x = torch.randn(8, 256, 256, W, device='cuda')
    c = ptwt_amd.wavedec3(x, 'db2', level=1, mode='zero')
    ms = t(lambda: ptwt_amd.waverec3(c, 'db2'))
    ma = t(lambda: ptwt_amd.wavedec3(x, 'db2', level=1, mode='zero'))
    print(f"W = {W}: waverec3 level 1 {ms:.4f} ms = {ms / W * 256:.4f} ms per 256 columns; wavedec3 level 1 {ma:.4f} ms = {ma / W * 256:.4f} per 256 columns", flush=True)
    del x, c
