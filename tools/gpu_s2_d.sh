#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 600 python tools/bench_more.py ) 2>/dev/null | tee gpurun_out/bench_more.log
( timeout 600 python bench.py --steps 5 --warmup 2 --workload fswavedec2_sym16_L5_32x8192x8192_f16 --no-cpu-baseline ) 2>&1 | tail -1 | cut -c1-1300 | tee gpurun_out/bench_c5.log
( timeout 600 python bench.py --steps 10 --warmup 2 --workload wavedec_db5_L10_32x1000000_f32 --no-cpu-baseline ) 2>&1 | tail -1 | cut -c1-1300 | tee gpurun_out/bench_1d.log
