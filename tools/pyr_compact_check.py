"""Kernel 16 in its compact form (MIFWT_OPT_DEBUG bit 11: eight waves per workgroup, two workgroups per CU, one loader wave) against the
default form: bit-identical coefficients on config 2 and a few odd shapes, then whole-call timing (results dropped / rotating)."""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
def run(x, wav, lev, mode, dbg):
    _engine.set_option(_engine.OPT_DEBUG, dbg)
    try:
        _engine.level_events = []
        c = ptwt_amd.wavedec2(x, wav, mode=mode, level=lev)
        torch.cuda.synchronize()
        kids = [e[1] for e in _engine.level_events]
    finally:
        _engine.level_events = None
        _engine.set_option(_engine.OPT_DEBUG, 0)
    return c, kids
torch.manual_seed(0)
for shape, wav, lev, mode in (((64, 1024, 1024), 'db4', 3, 'reflect'), ((3, 700, 1030), 'db2', 3, 'symmetric'), ((2, 515, 900), 'db3', 2, 'zero'),
                              ((5, 1024, 1024), 'haar', 3, 'constant'), ((2, 333, 1280), 'db4', 3, 'reflect'), ((4, 2048, 2048), 'db4', 3, 'reflect')):
    x = torch.randn(*shape, device='cuda')
    a, ka = run(x, wav, lev, mode, 0)
    b, kb = run(x, wav, lev, mode, 2048)
    same = torch.equal(a[0], b[0]) and all(torch.equal(u, v) for la, lb in zip(a[1:], b[1:]) for u, v in zip(la, lb))
    print(shape, wav, lev, mode, 'kernels', ka, kb, 'bit-identical' if same else 'DIFFERENT', flush=True)
