#!/bin/bash
# round 4: rolling strips with 10..16 taps — parity, then config 4 A/B; waverec2 C4 per-level breakdown
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pair_kernel" 2>&1 | tail -6 | tee gpurun_out/r04e_tests.txt
timeout 600 python -W ignore tools/roll16_ab.py 2>&1 | grep -v amdgpu | tee gpurun_out/r04e_roll16_ab.txt
timeout 300 python -W ignore tools/roll16_ab.py db5 5 32x1000x1000 2>&1 | grep -v amdgpu | tee -a gpurun_out/r04e_roll16_ab.txt
timeout 300 python bench.py --workload waverec2_db8_L4_64x4096x4096_f32 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04e_bench_waverec2_c4.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r04e_bench_waverec2_c4.json')); print(d['ms_per_step'], d['whole_call']['level_kernel_ms'], d['roofline']['kernel'])"
