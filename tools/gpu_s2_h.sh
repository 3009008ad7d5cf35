#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/pytest_gpu.log
( timeout 300 python tools/host_overhead.py ) 2>/dev/null | head -30
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['whole_call']['level_kernel_ms'], d['roofline']['achieved'])"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload wavedec_db5_L10_32x1000000_f32 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['whole_call']['level_kernel_ms'], d['roofline']['achieved'])"
