#!/bin/bash
export TMPDIR=/tmp
for w in wavedec2_db5_L5_32x1000x1000_f32_periodic fswavedec2_db5_L5_32x1000x1000_f32_periodic; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline --steps 200 > gpurun_out/r05r_bench_$w.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/r05r_bench_$w.json"))
print("$w", d["ms_per_step"], d["whole_call"]["frac_of_hbm_peak"], d["whole_call"]["level_kernel_ms"], d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"])
PY
done
