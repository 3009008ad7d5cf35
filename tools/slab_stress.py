"""Kernel 24, slab form: 300 launches of three geometries, every result compared bit for bit with the first (a race in the staged
slices / the filtered image would show as a run-to-run difference)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptwt_amd
from ptwt_amd import _engine as E
E.set_option(E.OPT_TILE_MODE, 4)
for shape, wav, mode in (((32, 100, 100, 100), 'db5', 'periodic'), ((16, 54, 54, 54), 'db4', 'reflect'), ((5, 31, 77, 128), 'db5', 'symmetric')):
    x = torch.randn(*shape, device='cuda')
    ref = ptwt_amd.wavedec3(x, wav, mode=mode, level=1)
    bad = 0
    for i in range(300):
        c = ptwt_amd.wavedec3(x, wav, mode=mode, level=1)
        if not (torch.equal(c[0], ref[0]) and all(torch.equal(c[1][k], ref[1][k]) for k in ref[1])): bad += 1
    print(shape, wav, mode, 'runs that differ from the first:', bad)
