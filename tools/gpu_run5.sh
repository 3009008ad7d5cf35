#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/pytest_gpu.log
( timeout 300 python tools/level_bench.py --inverse --shape 64,1024,1024 --rpc 0,8,16,32,64 --generic --rounds 3 ) 2>/dev/null | cut -c1-220
( timeout 300 python tools/level_bench.py --inverse --shape 64,515,515 --rpc 0,8,32 --rounds 3 ) 2>/dev/null | cut -c1-220
