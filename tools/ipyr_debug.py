"""Where does the streaming synthesis kernel (id 22) deviate from the oracle?  usage: ipyr_debug.py wavelet level H W [seg_rows]"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
from oracle import fwt_oracle as O
_engine.set_option(_engine.OPT_PYRAMID_MODE, 1)
def run(wav, level, H, W, seg=0, B=1):
    rng = np.random.default_rng(1)
    x = rng.standard_normal((B, H, W))
    c = O.wavedec2(x, wav, mode='reflect', level=level)
    c = (rng.standard_normal(c[0].shape),) + tuple(tuple(rng.standard_normal(b.shape) for b in lv) for lv in c[1:])
    conv = lambda t: torch.from_numpy(np.ascontiguousarray(t)).float().cuda()
    cd = tuple([conv(c[0])] + [tuple(conv(v) for v in lv) for lv in c[1:]])
    c32 = tuple([cd[0].cpu().double().numpy()] + [tuple(v.cpu().double().numpy() for v in lv) for lv in cd[1:]])
    if seg: _engine.set_option(_engine.OPT_PAIR_ROWS, seg)
    _engine.level_events = []
    got = ptwt_amd.waverec2(cd, wav).cpu().double().numpy()
    torch.cuda.synchronize()
    kids = [e[1] for e in _engine.level_events]; _engine.level_events = None
    _engine.set_option(_engine.OPT_PAIR_ROWS, 0)
    want = O.waverec2(c32, wav)
    err = np.abs(got - want)
    bad = err > 1e-4
    rows = np.where(bad.any(axis=(0, 2)))[0]; cols = np.where(bad.any(axis=(0, 1)))[0]
    print(f"{wav} L{level} {H}x{W} seg={seg} kids={kids} out={got.shape} coef widths {[lv[0].shape[-1] for lv in c[1:]]}: rel {np.linalg.norm(got-want)/np.linalg.norm(want):.2e}  bad rows {rows[:12]}..{rows[-3:] if len(rows) else ''} (n={len(rows)})  bad cols {cols[:12]}..{cols[-3:] if len(cols) else ''} (n={len(cols)})")
if len(sys.argv) > 1:
    run(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]) if len(sys.argv) > 5 else 0)
else:
    for wav in ('haar', 'db2'):
        for (H, W, lv) in ((203, 333, 3), (204, 334, 3), (204, 336, 3), (200, 334, 3), (203, 333, 2), (203, 333, 1), (300, 520, 3), (300, 522, 3), (64, 130, 1)):
            run(wav, lv, H, W)
