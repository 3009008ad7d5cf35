#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_autograd.py -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r04g_tests.txt
for w in wavedec2_bwd_db4_L3_64x1024x1024_f32 waverec2_bwd_db4_L3_64x1024x1024_f32; do
timeout 300 python bench.py --workload $w --steps 20 --warmup 5 > gpurun_out/r04g_bench_$w.json 2>gpurun_out/r04g_err.txt; python -c "
import json; d=json.load(open('gpurun_out/r04g_bench_$w.json')); print('$w', d['ms_per_step'], d['whole_call']['frac_of_hbm_peak'], d['whole_call']['level_kernel_ms'], d['roofline']['kernel'], d['roofline']['frac'], (d.get('cpu_baseline') or {}).get('value'))" || tail -5 gpurun_out/r04g_err.txt
done
