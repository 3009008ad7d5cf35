"""Where does the walking matrix-core kernel differ from the tile-at-a-time one?  Rows / columns of the first differences per band."""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
ptwt_amd.set_half_storage(True)
torch.manual_seed(0)
for shape, mode in [((1, 512, 512), 'zero'), ((1, 512, 512), 'reflect'), ((2, 300, 402), 'symmetric'), ((1, 1024, 2048), 'periodic'), ((3, 301, 403), 'reflect'), ((3, 333, 4111), 'constant'), ((5, 77, 95), 'periodic')]:
    x = torch.randn(*shape, device='cuda').half()
    _engine.set_option(7, 3)
    ref = ptwt_amd.wavedec2(x, 'sym16', mode=mode, level=1)
    _engine.set_option(7, 0)
    got = ptwt_amd.wavedec2(x, 'sym16', mode=mode, level=1)
    torch.cuda.synchronize()
    for name, a, b in zip(('aa', 'da', 'ad', 'dd'), [got[0], *got[1]], [ref[0], *ref[1]]):
        d = (a.float() - b.float()).abs()
        bad = d > 1e-2
        if not bad.any():
            print(shape, mode, name, 'equal' if torch.equal(a, b) else f'close (max {float(d.max()):.2e})')
            continue
        idx = bad.nonzero()
        rows = sorted(set(idx[:, -2].tolist())); cols = sorted(set(idx[:, -1].tolist()))
        def runs(v):
            out, s = [], v[0]
            for p, q in zip(v, v[1:] + [None]):
                if q != p + 1:
                    out.append((s, p)); s = q
            return out
        print(shape, mode, name, f'{int(bad.sum())} of {bad.numel()} wrong; rows {runs(rows)[:8]} cols {runs(cols)[:8]}')
