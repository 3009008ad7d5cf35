"""wavedec2 with and without the multi-level launch over plane sizes and level counts (where does kernel id 16 pay?)."""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
wav = sys.argv[1] if len(sys.argv) > 1 else 'db4'
def t(x, lev, pm):
    _engine.set_option(12, pm)
    for i in range(5): ptwt_amd.wavedec2(x, wav, level=lev)
    torch.cuda.synchronize()
    res = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(20): ptwt_amd.wavedec2(x, wav, level=lev)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
    _engine.set_option(12, 0)
    return sorted(res)[2]
for shape in [(256, 256, 256), (256, 512, 512), (64, 1024, 1024), (16, 2048, 2048), (64, 2048, 2048), (4, 4096, 4096), (64, 4096, 4096), (8, 1024, 1024), (1, 1024, 1024)]:
    x = torch.randn(*shape, device='cuda')
    print(shape, ' '.join(f"L{lev}: pyr {t(x, lev, 0):8.1f} other {t(x, lev, 2):8.1f} us |" for lev in (1, 2, 3)))
    del x
