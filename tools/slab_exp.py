"""Kernel 24, slab form: A/B runs of the experiment word (diagnostics build), 32 x 100^3 db5 periodic, one level."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptwt_amd as ptwt
from ptwt_amd import _engine as E
x = torch.randn(32, 100, 100, 100, device='cuda')
E.set_option(E.OPT_TILE_MODE, 4)
def t_us(reps=40):
    for _ in range(5): ptwt.wavedec3(x, 'db5', mode='periodic', level=1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ptwt.wavedec3(x, 'db5', mode='periodic', level=1)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for rep in range(2):
    for w in [int(v) for v in sys.argv[1:]] or [0, 1, 2, 3]:
        E.set_option(E.OPT_EXP, w)
        print('exp %d: %.1f us' % (w, t_us()))
