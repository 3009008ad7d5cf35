"""The matrix-core synthesis kernel (id 23, forced with MIFWT_OPT_MFMA_MODE 4) against the vector tile kernel (mode 2) and the fp64 oracle."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
from oracle import fwt_oracle as O
ptwt_amd.set_half_storage(True)
rng = np.random.default_rng(0)
def relerr(a, b): return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
for shape, wav, mode in [((1, 512, 512), 'sym16', 'zero'), ((2, 300, 402), 'sym16', 'symmetric'), ((1, 131, 259), 'db10', 'reflect'), ((3, 333, 1031), 'db12', 'periodic'),
                         ((2, 96, 200), 'db9', 'constant'), ((1, 1024, 2048), 'db14', 'reflect'), ((5, 77, 95), 'sym16', 'periodic')]:
    x = rng.standard_normal(shape)
    try:
        c64 = O.wavedec2(x, wav, mode=mode, level=1)
    except RuntimeError as e:
        print(shape, wav, mode, 'oracle refuses:', e); continue
    # random coefficients of those shapes, f16-quantised
    cq = [torch.from_numpy(rng.standard_normal(c64[0].shape)).half()] + [tuple(torch.from_numpy(rng.standard_normal(b.shape)).half() for b in c64[1])]
    cdev = (cq[0].cuda(), tuple(t.cuda() for t in cq[1]))
    want = O.waverec2((cq[0].double().numpy(), tuple(t.double().numpy() for t in cq[1])), wav)
    out = {}
    for m in (2, 4):
        _engine.set_option(7, m)
        _engine.level_events = []
        y = ptwt_amd.waverec2(cdev, wav); torch.cuda.synchronize()
        kids = [e[1] for e in _engine.level_events]; _engine.level_events = None
        out[m] = (y.double().cpu().numpy(), kids)
    _engine.set_option(7, 0)
    e2, e4 = relerr(out[2][0], want), relerr(out[4][0], want)
    d = np.abs(out[4][0] - want)
    bad = np.argwhere(d > 0.05 * max(1.0, np.abs(want).max()))
    msg = '' if len(bad) == 0 else f'  BAD {len(bad)}: rows {bad[:,-2].min()}..{bad[:,-2].max()} cols {bad[:,-1].min()}..{bad[:,-1].max()}'
    print(f"{shape} {wav} {mode}: out {out[4][0].shape} kernels {out[2][1]} / {out[4][1]}  rel err vector {e2:.2e}  matrix {e4:.2e}{msg}", flush=True)
