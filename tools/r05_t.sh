#!/bin/bash
export TMPDIR=/tmp
timeout 900 python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r05t_bench.json 2>/dev/null
python -c "
import json
d=json.load(open('gpurun_out/r05t_bench.json'))
print(d['ms_per_step'], d['whole_call']['frac_of_hbm_peak'], 'rot', d['whole_call']['rotating_outputs_ms'], d['whole_call']['rotating_outputs_frac_of_hbm_peak'], 'kernel', d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
