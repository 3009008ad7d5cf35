#!/bin/bash
export TMPDIR=/tmp
( timeout 200 python bench.py > gpurun_out/r04_final8_bench_default.json 2> gpurun_out/r04_final_bench.err )
python -c "
import json
d=json.load(open('gpurun_out/r04_final8_bench_default.json')); print(d['ms_per_step'], d['whole_call']['frac_of_hbm_peak'], d['whole_call']['rotating_outputs_ms'], d['roofline']['frac'], d['roofline']['consistent']); [print(s['workload'], s.get('ms_per_step'), s.get('frac')) for s in d['secondary']]"
