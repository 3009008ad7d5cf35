#!/bin/bash
# round 3, call: bench lines (waverec2 + headline), rocprof stats + PMC of the streaming synthesis kernel
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 400 python bench.py --workload waverec2_db4_L3_64x1024x1024_f32 ) 2>/dev/null | tail -1 > gpurun_out/r03d_bench_waverec2_db4_L3_64x1024x1024_f32.json
python -c "import json; d=json.load(open('gpurun_out/r03d_bench_waverec2_db4_L3_64x1024x1024_f32.json')); print(d['ms_per_step'], d['value'], d['whole_call']['frac_of_hbm_peak'], d['roofline'], d.get('cpu_baseline'))"
( timeout 400 python bench.py ) 2>/dev/null | tail -1 > gpurun_out/r03d_bench_config2.json
python -c "import json; d=json.load(open('gpurun_out/r03d_bench_config2.json')); print(d['ms_per_step'], d['value'], d['whole_call']['frac_of_hbm_peak'], d['roofline']['frac'], d.get('cpu_baseline',{}).get('value'))"
TAG=r03d WL=waverec2_db4_L3_64x1024x1024_f32 KERNEL=idwt2_pyr_kernel bash tools/pmc_workload.sh 2>&1 | tail -60
