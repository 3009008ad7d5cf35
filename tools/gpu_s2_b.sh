#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -30 gpurun_out/pytest_gpu.log
( timeout 300 python bench.py --steps 20 --warmup 3 --workload wavedec3_db2_L3_8x256x256x256_f32 --no-cpu-baseline ) > gpurun_out/bench_c3.log 2>&1
tail -1 gpurun_out/bench_c3.log | cut -c1-1200
