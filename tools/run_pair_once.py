#!/usr/bin/env python
"""One config-2 wavedec2 per variant (pair kernel / per-level kernels), a few repetitions: the workload under tools/pmc_any.sh.
MIFWT_PAIR_MODES="0,1,2" selects the variants (OPT_PAIR_MODE values), MIFWT_PAIR_ROWS the OPT_PAIR_ROWS override."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ptwt_amd  # noqa: E402
from ptwt_amd import _engine  # noqa: E402

shape = tuple(int(v) for v in os.environ.get("MIFWT_SHAPE", "64,1024,1024").split(","))
wavelet = os.environ.get("MIFWT_WAVELET", "db4")
level = int(os.environ.get("MIFWT_LEVEL", "3"))
x = [torch.randn(*shape, device="cuda") for _ in range(3)]
_engine.set_option(_engine.OPT_PAIR_ROWS, int(os.environ.get("MIFWT_PAIR_ROWS", "0")))
for rep in range(4):
    for pm in [int(v) for v in os.environ.get("MIFWT_PAIR_MODES", "0,2").split(",")]:
        _engine.set_option(_engine.OPT_PAIR_MODE, pm)
        ptwt_amd.wavedec2(x[rep % 3], wavelet, level=level)
torch.cuda.synchronize()
