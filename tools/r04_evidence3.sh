#!/bin/bash
# round 4, evidence for the 3-D walk kernels (ids 24 / 25) + the final suite / smoke / default bench of the round
export TMPDIR=/tmp
TAG=r04z2
run_pmc() { TAG=$TAG WL=$1 KERNEL=$2 STEPS=$3 bash tools/pmc_workload.sh > gpurun_out/${TAG}_pmc_$1.log 2>&1; }
run_pmc wavedec3_db2_L3_8x256x256x256_f32 dwt3_fwd_walk_kernel 30
run_pmc waverec3_db2_L3_8x256x256x256_f32 idwt3_walk_kernel 30
cp gpurun_out/${TAG}_pmc_*.json profiles/ 2>/dev/null
for wl in wavedec3_db2_L3_8x256x256x256_f32 waverec3_db2_L3_8x256x256x256_f32; do
  ( timeout 600 python bench.py --workload $wl --steps 100 --warmup 10 ) 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_$wl.json
  python -c "
import json
d=json.load(open('gpurun_out/${TAG}_bench_$wl.json')); r=d['roofline']
print('$wl', 'ms/step', d['ms_per_step'], 'whole', d['whole_call']['frac_of_hbm_peak'], 'rot', d['whole_call']['rotating_outputs_ms'], d['whole_call']['level_kernel_ms'], r['kernel'], r['avg_launch_ms'], r['frac'], 'traffic', r['traffic'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
done
( time timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -5 ) 2>&1 | tee gpurun_out/r04_final_gpu_suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a gpurun_out/r04_final_gpu_suite.txt
( time timeout 600 python bench.py > gpurun_out/r04_final6_bench_default.json 2> gpurun_out/r04_final_bench.err ) 2>&1 | tail -3
python -c "
import json
d=json.load(open(\"gpurun_out/r04_final6_bench_default.json\")); print(d['ms_per_step'], d['whole_call']['frac_of_hbm_peak'], d['whole_call']['rotating_outputs_ms'], d['roofline']['frac'], d['roofline']['consistent']); [print(s) for s in d['secondary']]"
