#!/bin/bash
export TMPDIR=/tmp
( time timeout 600 python bench.py > gpurun_out/r04_final5_bench_default.json 2> gpurun_out/r04_final5_bench.err ) 2>&1 | tail -3
python -c "
import json
d=json.load(open(\"gpurun_out/r04_final5_bench_default.json\")); print(d['ms_per_step'], d['whole_call']['frac_of_hbm_peak'], d['whole_call']['rotating_outputs_ms'], d['roofline']['frac'], d['roofline']['consistent'], d['roofline'].get('avg_launch_ms'))"
timeout 300 python tools/c3_opt_sweep.py 2>&1 | tee gpurun_out/r04v_c3_opt_sweep.txt
