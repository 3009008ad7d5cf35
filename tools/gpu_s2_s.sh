#!/bin/bash
export TMPDIR=/tmp
python - <<'PY'
import torch, numpy as np, sys, json
sys.path.insert(0, '.')
import __graft_entry__ as g; g.build(verbose=False)
import ptwt_amd
from ptwt_amd import _engine
from oracle import fwt_oracle as O
dev = torch.device('cuda:0')
ptwt_amd.set_half_storage(True)
rng = np.random.default_rng(0)
for wavelet in ['sym16', 'db9', 'db12']:
    flen = len(O.filter_bank(wavelet)[0])
    for shape in [(2, 131, 3 * flen + 70), (1, 300, 401)]:
        xq = torch.from_numpy(rng.standard_normal(shape)).half()
        for mode in ['reflect', 'zero', 'symmetric', 'periodic', 'constant']:
            want = O.wavedec2(xq.double().numpy(), wavelet, mode=mode, level=1)
            res = {}
            for name, opt in (('mfma', 0), ('vector', 2)):
                _engine.set_option(7, opt)
                kid = _engine.kernel_id(2, torch.float16, mode, flen, shape[0], shape[1:])
                got = ptwt_amd.wavedec2(xq.to(dev), wavelet, mode=mode, level=1)
                errs = [float(np.linalg.norm(a.cpu().double().numpy() - b) / np.linalg.norm(b)) for a, b in zip([got[0], *got[1]], [want[0], *want[1]])]
                res[name] = (kid, max(errs))
            print(wavelet, shape, mode, {k: (v[0], '%.2e' % v[1]) for k, v in res.items()})
_engine.set_option(7, 0)
# timing on the config-5 level-1 shape (8 images)
x = [torch.randn(8, 8192, 8192, device=dev).half() for _ in range(3)]
for name, opt in (('mfma', 0), ('vector', 2)):
    _engine.set_option(7, opt)
    for i in range(3): ptwt_amd.wavedec2(x[i], 'sym16', level=1)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(6): ptwt_amd.wavedec2(x[i % 3], 'sym16', level=1)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 6
    bytes_ = 8 * (8192 * 8192 + 4 * 4111 * 4111) * 2
    print(name, 'level-1 8x8192^2 f16 sym16:', round(ms, 3), 'ms', round(bytes_ / ms / 1e6, 1), 'GB/s')
_engine.set_option(7, 0)
PY
