"""config 3 (8 x 256^3 db2 level 3): row pitch of the sub-band planes (MIFWT_ROW_ALIGN bytes; set in the environment)."""
import os, sys, torch
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
x = torch.randn(8, 256, 256, 256, device='cuda')
c = ptwt_amd.wavedec3(x, 'db2', level=3, mode='zero')
print(f"MIFWT_ROW_ALIGN={os.environ.get('MIFWT_ROW_ALIGN', '1')}: band pitch {c[-1]['ddd'].stride(-2)} floats; wavedec3 {t(lambda: ptwt_amd.wavedec3(x, 'db2', level=3, mode='zero')):.4f} ms   waverec3 {t(lambda: ptwt_amd.waverec3(c, 'db2')):.4f} ms")
