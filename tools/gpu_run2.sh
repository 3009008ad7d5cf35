#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/pytest_gpu.log
( timeout 300 python tools/level_bench.py --rpc 8,16,32,64 --depth 1,2,3 --rounds 3 ) > gpurun_out/level_bench.log 2>&1; echo "level_bench rc=$?"
cat gpurun_out/level_bench.log
( timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline ) > gpurun_out/bench.log 2>&1; echo "bench rc=$?"
tail -1 gpurun_out/bench.log
