#!/bin/bash
export TMPDIR=/tmp
python - <<'PY'
import torch, sys, json
sys.path.insert(0, '.')
import __graft_entry__ as g; g.build(verbose=False)
import ptwt_amd
from ptwt_amd import _engine
dev = torch.device('cuda:0')
xs = [torch.randn(8, 256, 256, 256, device=dev) for _ in range(3)]
ref = None
for name, opt5, opt6 in [('brick TD2 TR4', 0, 3), ('brick TD2 TR2', 0, 4), ('brick TD2 TR8', 0, 5), ('composed', 2, 0)]:
    _engine.set_option(5, opt5); _engine.set_option(6, opt6)
    c = ptwt_amd.wavedec3(xs[0], 'db2', level=1)
    if ref is None: ref = c
    else:
        print(name, 'max diff vs first', max(float((c[1][k] - ref[1][k]).abs().max()) for k in c[1]))
    for lvl in (1, 3):
        for i in range(3): ptwt_amd.wavedec3(xs[i], 'db2', level=lvl)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(12): ptwt_amd.wavedec3(xs[i % 3], 'db2', level=lvl)
        e.record(); torch.cuda.synchronize()
        print(name, 'levels', lvl, round(s.elapsed_time(e) / 12, 4), 'ms')
_engine.set_option(5, 0); _engine.set_option(6, 0)
PY
