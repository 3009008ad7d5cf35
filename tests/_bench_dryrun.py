"""TEST-ONLY launcher of ``bench.py`` without a GPU: the level engine is replaced by the tests' numpy stand-in and bench.py is
told to keep its tensors on the host (``MIFWT_BENCH_DEVICE=cpu``).  Run by tests/test_bench_dryrun.py, alone and under
``python -m torch.distributed.run`` with the gloo backend, to cover argument handling, rendezvous, barriers, the
max-over-ranks reduction and the JSON line of the N > 1 path before a multi-GPU node ever sees it."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MIFWT_BENCH_DEVICE"] = "cpu"

import ptwt_amd  # noqa: E402,F401
from ptwt_amd import _engine  # noqa: E402
from tests._oracle_engine import OracleLevelEngine  # noqa: E402

_engine.ENGINE = OracleLevelEngine()
sys.argv[0] = os.path.join(ROOT, "bench.py")
runpy.run_path(sys.argv[0], run_name="__main__")
