#!/bin/bash
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -5 ) 2>&1 | tee gpurun_out/r04_final_gpu_suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a gpurun_out/r04_final_gpu_suite.txt
( time timeout 600 python bench.py > gpurun_out/r04_final4_bench_default.json 2> gpurun_out/r04_final_bench.err ) 2>&1 | tail -3
python -c "
import json
d=json.load(open(\"gpurun_out/r04_final4_bench_default.json\")); print(d['ms_per_step'], d['whole_call']['frac_of_hbm_peak'], d['whole_call']['rotating_outputs_ms'], d['roofline']['frac'], d['roofline']['consistent']); [print(s) for s in d['secondary']]"
