#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/pytest_gpu.log
( timeout 300 python bench.py --steps 50 --warmup 5 ) > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log | cut -c1-1400
( timeout 300 python bench.py --steps 10 --warmup 3 --workload wavedec2_db8_L4_64x4096x4096_f32 --no-cpu-baseline ) 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['whole_call'])"
( timeout 600 python tools/bench_more.py config2 config3 ) 2>/dev/null
