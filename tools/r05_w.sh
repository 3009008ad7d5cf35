#!/bin/bash
# f64 volumes on the walk kernels: parity, canaries, routes, the bench workloads, ten taps
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests/test_gpu_walk3.py tests/test_gpu_canaries.py "tests/test_gpu_parity.py::test_stream_routes_selected" tests/test_gpu_parity.py -x -q 2>&1 | tail -6
for w in wavedec3_db2_L3_8x256x256x256_f64 waverec3_db2_L3_8x256x256x256_f64; do
  timeout 600 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > gpurun_out/r05w_bench_$w.json
  python - <<PY
import json
d=json.loads(open('gpurun_out/r05w_bench_$w.json').read())
print('$w', d['ms_per_step'], d['roofline'].get('frac'), d['roofline'].get('frac_unchecked'), d.get('whole_call'))
PY
done
timeout 600 python -W ignore tools/walk3_f64.py db5 2>&1 | grep -v amdgpu
} 2>&1 | tee gpurun_out/r05w_f64_walk.txt
