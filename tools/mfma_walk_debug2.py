import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
ptwt_amd.set_half_storage(True)
mode = sys.argv[1] if len(sys.argv) > 1 else 'zero'
bad = []
for r0 in range(0, 6):
    for c0 in range(0, 6):
        x = torch.zeros(1, 512, 512, device='cuda').half()
        x[0, r0, c0] = 1.0
        _engine.set_option(7, 3); ref = ptwt_amd.wavedec2(x, 'sym16', mode=mode, level=1)
        _engine.set_option(7, 0); got = ptwt_amd.wavedec2(x, 'sym16', mode=mode, level=1)
        d = (got[0].float() - ref[0].float()).abs()
        if float(d.max()) > 1e-4:
            idx = (d > 1e-4).nonzero()
            bad.append((r0, c0, float(d.max()), idx[:, -2].min().item(), idx[:, -2].max().item(), idx[:, -1].min().item(), idx[:, -1].max().item()))
print(mode, 'input samples whose response differs (r, c, max diff, out rows lo..hi, out cols lo..hi):')
for b in bad: print(' ', b)
# which value does the walk kernel "see" there?  response of a constant plane
