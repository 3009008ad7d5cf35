#!/bin/bash
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_ipyr.py -q -m gpu -x 2>&1 | tail -4
timeout 600 python bench.py --workload waverec2_db4_L3_64x1024x1024_f32 --no-cpu-baseline --no-secondary > gpurun_out/r05v_bench_waverec2.json 2>/dev/null
python -c "
import json
d=json.load(open('gpurun_out/r05v_bench_waverec2.json'))
print('waverec2', d['ms_per_step'], d['whole_call']['frac_of_hbm_peak'], 'rot', d['whole_call']['rotating_outputs_ms'], 'kernel', d['roofline']['frac_unchecked'], d['roofline']['avg_launch_ms'])"
