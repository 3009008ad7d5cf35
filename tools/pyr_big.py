"""64 x 4096^2 wavedec2 db4 L3 through the multi-level launch: row segments / padded row pitch (memory-channel conflicts?)"""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
def t(x, lev=3):
    for i in range(3): ptwt_amd.wavedec2(x, 'db4', level=lev)
    torch.cuda.synchronize()
    res = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(10): ptwt_amd.wavedec2(x, 'db4', level=lev)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 10 * 1e3)
    return sorted(res)[1]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x = torch.randn(B, 4096, 4096, device='cuda')
print('dense, auto segments', t(x))
for sr in (64, 128, 256):
    _engine.set_option(_engine.OPT_PAIR_ROWS, sr)
    print('dense, segment rows', sr, t(x))
_engine.set_option(_engine.OPT_PAIR_ROWS, 0)
big = torch.randn(B, 4096, 4096 + 64, device='cuda')
print('row pitch 4160, auto segments', t(big[:, :, :4096]))
big = torch.randn(B, 4096 + 8, 4096 + 64, device='cuda')
print('row pitch 4160, image pitch 4104 rows', t(big[:, :4096, :4096]))
_engine.set_option(12, 2)
print('other kernels, dense', t(x))
