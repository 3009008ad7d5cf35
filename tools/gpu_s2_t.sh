#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/pytest_gpu.log
( timeout 600 python bench.py --steps 10 --warmup 3 --workload fswavedec2_sym16_L5_32x8192x8192_f16 --no-cpu-baseline ) 2>/dev/null | tail -1 > gpurun_out/bench_fswavedec2_sym16_L5_32x8192x8192_f16.log
python -c "import json; d=json.loads(open('gpurun_out/bench_fswavedec2_sym16_L5_32x8192x8192_f16.log').read()); print(d['ms_per_step'], d['value'], d['whole_call'], d['roofline']['kernel'], d['roofline']['frac'])"
