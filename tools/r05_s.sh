#!/bin/bash
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_walk3.py tests/test_gpu_parity.py tests/test_gpu_canaries.py -q -m gpu -x 2>&1 | tail -6 | tee gpurun_out/r05s_tests.txt
for w in wavedec3_db2_L3_8x256x256x256_f64; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline --no-secondary --steps 50 > gpurun_out/r05s_bench_$w.json 2>/dev/null
  python -c "
import json
d=json.load(open('gpurun_out/r05s_bench_$w.json'))
print('$w', d['ms_per_step'], d['whole_call']['frac_of_hbm_peak'], d['whole_call']['level_kernel_ms'], d['roofline']['kernel'])"
done
