#!/bin/bash
export TMPDIR=/tmp
run() { ( timeout 300 python tools/level_bench.py "$@" --rounds 3 --iters 10 ) 2>/dev/null | cut -c1-175; }
python - <<'PY'
import torch, numpy as np, sys
sys.path.insert(0, '.')
import __graft_entry__ as g; g.build(verbose=False)
import ptwt_amd
from ptwt_amd import _engine
from oracle import fwt_oracle as O
dev = torch.device('cuda:0')
rng = np.random.default_rng(0)
bad = 0
for wavelet in ['haar', 'db2', 'db4', 'db8', 'db10', 'sym16']:
    for shape in [(3, 70, 530), (2, 131, 257), (2, 40, 36)]:
        x = rng.standard_normal(shape)
        flen = len(O.filter_bank(wavelet)[0])
        level = 2 if min(shape[1:]) > 4 * flen else 1
        c64 = O.wavedec2(x, wavelet, mode='symmetric', level=level)
        want = O.waverec2(c64, wavelet)
        cg = (torch.from_numpy(c64[0]).float().to(dev),) + tuple(tuple(torch.from_numpy(t).float().to(dev) for t in det) for det in c64[1:])
        for tro in (0, 8, 16, 24, 32):
            _engine.set_option(5, 1); _engine.set_option(6, tro)
            got = ptwt_amd.waverec2(cg, wavelet).cpu().double().numpy()
            err = np.linalg.norm(got - want) / np.linalg.norm(want)
            if err > 1e-6 or got.shape != want.shape:
                bad += 1; print('BAD', wavelet, shape, tro, err)
_engine.set_option(5, 0); _engine.set_option(6, 0)
print('idwt tile check: bad =', bad)
PY
run --inverse --shape 64,1024,1024 --tile 2,1 --tr 0,8,16,24,32
run --inverse --shape 64,515,515 --tile 2,1 --tr 0,16
run --inverse --shape 64,261,261 --tile 2,1 --tr 0,16
run --inverse --shape 64,4096,4096 --wavelet db8 --tile 2,1 --tr 16,32
run --inverse --shape 64,1024,1024 --wavelet haar --tile 2,1 --tr 16
