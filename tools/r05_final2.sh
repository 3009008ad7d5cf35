#!/bin/bash
# the round's last evidence with the final library: full GPU suite, smoke(), PMC + kernel stats + bench lines of the workloads whose kernels
# changed after r05z (f64 volumes on the walk kernels, forward + backward, the matrix-core walk), the default bench run
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/r05zz_gpu_suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/r05zz_smoke.txt
WLS="wavedec3_db2_L3_8x256x256x256_f64:dwt3_fwd_walk_kernel:20 waverec3_db2_L3_8x256x256x256_f64:idwt3_walk_kernel:20 wavedec2_bwd_db4_L3_64x1024x1024_f32:dwt2_fwd_pyr_kernel:40 waverec2_bwd_db4_L3_64x1024x1024_f32:idwt2_pyr_kernel:40" TAG=r05zz bash tools/r05_evidence.sh > gpurun_out/r05zz_evidence.log 2>&1
grep -v "^  File\|^Traceback\|^    \|json.decoder" gpurun_out/r05zz_evidence.log | tail -10 | tee gpurun_out/r05zz_evidence_summary.txt
timeout 900 python bench.py > gpurun_out/r05zz_bench_default_run.json 2> gpurun_out/r05zz_bench_default_err.txt
python - <<PY
import json
d=json.loads(open("gpurun_out/r05zz_bench_default_run.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["roofline"]["frac"], d["whole_call"].get("rotating_outputs_ms"))
for s in d.get("secondary", []): print(s.get("workload"), s.get("ms_per_step"), s.get("frac"))
PY
