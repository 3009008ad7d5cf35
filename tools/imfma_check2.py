import sys, numpy as np, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
from oracle import fwt_oracle as O
ptwt_amd.set_half_storage(True)
_engine.set_option(7, 4)
def relerr(a, b): return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
for wavelet in ("db9", "sym16"):
    rng = np.random.default_rng(len(wavelet) + 300)
    flen = len(O.filter_bank(wavelet)[0])
    for shape, seg in [((2, 131, 3 * flen + 70), 0), ((1, 2 * flen, 2 * flen + 1), 0), ((3, 300, 402), 3), ((40, 96, 200), 1), ((1, 700, 1031), 5)]:
        for mode in ("reflect", "periodic", "zero"):
            try:
                c64 = O.wavedec2(rng.standard_normal(shape), wavelet, mode=mode, level=1)
            except RuntimeError:
                continue
            cq = [torch.from_numpy(rng.standard_normal(c64[0].shape)).half()] + [tuple(torch.from_numpy(rng.standard_normal(b.shape)).half() for b in c64[1])]
            want = O.waverec2((cq[0].double().numpy(), tuple(t.double().numpy() for t in cq[1])), wavelet)
            cdev = (cq[0].cuda(), tuple(t.cuda() for t in cq[1]))
            for sg in sorted({0, seg}):
                _engine.set_option(6, sg)
                got = ptwt_amd.waverec2(cdev, wavelet).double().cpu().numpy()
                _engine.set_option(6, 0)
                e = relerr(got, want)
                msg = ''
                if e > 5e-4:
                    d = np.abs(got - want); bad = np.argwhere(d > 0.02 * np.abs(want).max())
                    msg = f'  BAD {len(bad)}: imgs {sorted(set(bad[:,0].tolist()))[:5]} rows {bad[:,-2].min()}..{bad[:,-2].max()} cols {bad[:,-1].min()}..{bad[:,-1].max()}'
                print(f"{wavelet} {shape} coef {tuple(c64[0].shape)} {mode} seg {sg}: {e:.2e}{msg}", flush=True)
