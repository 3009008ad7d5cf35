"""A few level-1 wavedec3 calls on BASELINE config 3 (8 x 256^3 db2): the workload under tools/pmc_any.sh (KERNEL=dwt3_fwd)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptwt_amd
from ptwt_amd import _engine
xs = [torch.randn(8, 256, 256, 256, device="cuda") for _ in range(3)]
_engine.set_option(6, int(os.environ.get("MIFWT_T6", "0")))
for i in range(5):
    ptwt_amd.wavedec3(xs[i % 3], os.environ.get("MIFWT_WAVELET", "db2"), level=1)
torch.cuda.synchronize()
