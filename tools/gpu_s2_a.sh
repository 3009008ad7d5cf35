#!/bin/bash
# session-2 baseline: parity tests, bench lines, host overhead, PMC passes + kernel trace
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/pytest_gpu.log
( timeout 300 python bench.py --steps 50 --warmup 5 ) > gpurun_out/bench.log 2>&1; echo "bench rc=$?"
tail -1 gpurun_out/bench.log | cut -c1-1500
( timeout 300 python bench.py --steps 20 --warmup 3 --workload wavedec3_db2_L3_8x256x256x256_f32 --no-cpu-baseline ) > gpurun_out/bench_c3.log 2>&1
tail -1 gpurun_out/bench_c3.log | cut -c1-300
( timeout 300 python tools/host_overhead.py ) > gpurun_out/host_overhead.log 2>&1
head -40 gpurun_out/host_overhead.log
RPC=0 DEPTH=0 bash tools/gpu_pmc.sh
