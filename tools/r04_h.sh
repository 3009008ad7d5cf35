#!/bin/bash
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -8 ) 2>&1 | tee gpurun_out/r04h_full_gpu_suite.txt
for w in wavedec2_bwd_db4_L3_64x1024x1024_f32; do
timeout 300 python bench.py --workload $w --steps 20 --warmup 5 > gpurun_out/r04h_bench_$w.json 2>gpurun_out/r04h_err.txt; python -c "
import json; d=json.load(open('gpurun_out/r04h_bench_$w.json')); print('$w', d['ms_per_step'], d['whole_call']['frac_of_hbm_peak'], d['whole_call']['level_kernel_ms'])" || tail -5 gpurun_out/r04h_err.txt
done
