#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/pytest_gpu.log
( timeout 600 python bench.py --steps 5 --warmup 2 --workload fswavedec2_sym16_L5_32x8192x8192_f16 --no-cpu-baseline ) 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['whole_call'], d['roofline']['achieved'])"
( timeout 600 python tools/bench_more.py sym16 config3 ) 2>/dev/null
