"""Kernel 24, slab form (32 x N^3 db5 periodic, one level): what the launch costs with parts of it switched off (diagnostics build:
MIFWT_LIB=libmifwt_diag.so).  Bits: 1 no stores, 2 no requests, 4 no row pass, 8 no column / depth pass, 16 no pad fill."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptwt_amd as ptwt
from ptwt_amd import _engine as E
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
wav = sys.argv[2] if len(sys.argv) > 2 else 'db5'
x = torch.randn(32, n, n, n, device='cuda')
E.set_option(E.OPT_TILE_MODE, 4)
def t_us(reps=30):
    for _ in range(5): ptwt.wavedec3(x, wav, mode='periodic', level=1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ptwt.wavedec3(x, wav, mode='periodic', level=1)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for name, bits in [('all on', 0), ('no stores', 1), ('no requests', 2), ('no pad fill', 16), ('no requests, no pad fill (loaders idle)', 18), ('no row pass', 4),
                   ('no column / depth pass', 8), ('no row / column / depth pass, no stores (loaders only)', 13), ('compute only (no requests / pad fill / stores)', 19),
                   ('barriers only', 31), ('loaders: barriers alone', 31 + 32), ('compute: no depth pass either', 31 + 64), ('nothing but barriers', 127), ('row pass alone (compute waves)', 1 + 8 + 64 + 32), ('column + depth pass alone', 1 + 4 + 32), ('row + column + depth pass, loaders off', 1 + 32), ('the same + stores', 32)]:
    E.set_option(E.OPT_DEBUG, bits)
    print('%-70s %.1f us' % (name, t_us()))
