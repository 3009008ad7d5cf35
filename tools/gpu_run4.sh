#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/pytest_gpu.log
( timeout 300 python tools/level_bench.py --shape 64,1024,1024 --coop 1,0 --rpc 0,16,32,34,64 --depth 1,2 --rounds 3 ) 2>/dev/null | cut -c1-200
( timeout 300 python tools/level_bench.py --shape 64,515,515 --coop 1,0 --rpc 0,16,32 --depth 1 --rounds 3 ) 2>/dev/null | cut -c1-200
( timeout 300 python tools/level_bench.py --shape 64,261,261 --coop 1,0 --rpc 0,16,32 --depth 1 --rounds 3 ) 2>/dev/null | cut -c1-200
( timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline ) > gpurun_out/bench.log 2>&1; echo "bench rc=$?"
tail -1 gpurun_out/bench.log | cut -c1-1200
