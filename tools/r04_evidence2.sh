#!/bin/bash
# round 4, evidence refresh for the workloads whose kernels changed after tools/r04_evidence.sh ran (border kernel, 32-row synthesis tasks)
export TMPDIR=/tmp
TAG=r04z
run_pmc() { TAG=$TAG WL=$1 KERNEL=$2 STEPS=$3 KERNEL2=$4 bash tools/pmc_workload.sh > gpurun_out/${TAG}_pmc_$1.log 2>&1; }
run_pmc wavedec2_bwd_db4_L3_64x1024x1024_f32 dwt2_fwd_pyr_kernel 40
run_pmc waverec2_db8_L4_64x4096x4096_f32 dwt2_inv_stream_kernel 10
for wl in wavedec2_bwd_db4_L3_64x1024x1024_f32 waverec2_bwd_db4_L3_64x1024x1024_f32 waverec2_db8_L4_64x4096x4096_f32; do
  steps=100; case $wl in *4096x4096*) steps=20;; esac
  ( timeout 600 python bench.py --workload $wl --steps $steps --warmup 10 ) 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_$wl.json
  python -c "
import json
d=json.load(open('gpurun_out/${TAG}_bench_$wl.json')); r=d['roofline']
print('$wl', 'ms/step', d['ms_per_step'], 'whole', d['whole_call']['frac_of_hbm_peak'], d['whole_call']['level_kernel_ms'], 'traffic', r['traffic'])"
done
( timeout 600 python bench.py ) 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_default_run.json
python -c "
import json
d=json.load(open('gpurun_out/${TAG}_bench_default_run.json')); print(d['ms_per_step'], d['whole_call']['rotating_outputs_ms'], [(s['workload'], s.get('ms_per_step'), s.get('frac')) for s in d['secondary']])"
