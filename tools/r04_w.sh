#!/bin/bash
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_pyramid.py tests/test_gpu_ipyr.py tests/test_gpu_walk3.py -q 2>&1 | tail -4 ) 2>&1 | tee gpurun_out/r04x4_final_subset.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a gpurun_out/r04x4_final_subset.txt
( timeout 600 python bench.py > gpurun_out/r04_final7_bench_default.json 2> gpurun_out/r04_final_bench.err )
python -c "
import json
d=json.load(open('gpurun_out/r04_final7_bench_default.json')); print(d['ms_per_step'], d['whole_call']['frac_of_hbm_peak'], d['whole_call']['rotating_outputs_ms'], d['roofline']['frac'], d['roofline']['consistent']); [print(s['workload'], s.get('ms_per_step'), s.get('frac')) for s in d['secondary']]"
