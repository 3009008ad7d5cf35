"""Levels of the reference's 3-D speed shape (32 x 100^3 db5 periodic): composed route (planes + depth pass) against the depth-walking
kernel (MIFWT_OPT_TILE_MODE 4), one level per call, us."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptwt_amd as ptwt
from ptwt_amd import _engine as E
def timeit(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for wav in ('db5', 'db4', 'db2'):
    cur = torch.randn(32, 100, 100, 100, device='cuda')
    for l in range(3):
        row = []
        for tm in (0, 4):
            E.set_option(E.OPT_TILE_MODE, tm)
            row.append(timeit(lambda: ptwt.wavedec3(cur, wav, mode='periodic', level=1)))
        E.set_option(E.OPT_TILE_MODE, 0)
        c = ptwt.wavedec3(cur, wav, mode='periodic', level=1)
        print('%s level %d %s: default %.1f us, walk %.1f us' % (wav, l + 1, tuple(cur.shape[1:]), row[0], row[1]))
        cur = c[0].contiguous()
