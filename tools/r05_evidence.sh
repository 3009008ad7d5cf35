#!/bin/bash
# round-5 evidence run: PMC + kernel-trace summaries per workload (dominant kernel), then one bench line per workload that carries
# roofline + traffic (from the PMC summary just taken) + cpu_baseline.  Everything lands under gpurun_out/ (r05y_*); copy to profiles/.
export TMPDIR=/tmp
mkdir -p gpurun_out
TAG=${TAG:-r05y}
run_pmc() {  # workload, kernel substring, steps
  [ -n "$SKIP_PMC" ] && return
  TAG=$TAG WL=$1 KERNEL=$2 STEPS=$3 KERNEL2=$4 bash tools/pmc_workload.sh > gpurun_out/${TAG}_pmc_$1.log 2>&1
  cp gpurun_out/${TAG}_pmc_$1.json profiles/ 2>/dev/null
  python -c "import json; d=json.load(open('gpurun_out/${TAG}_pmc_$1.json')); print('$1', d.get('kernel','?')[:60], 'traffic', d.get('hbm_traffic_bytes'), d.get('launch_ns_by_grid'))"
}
WLS=${WLS:-"wavedec2_db4_L3_64x1024x1024_f32:dwt2_fwd_pyr_kernel:60 waverec2_db4_L3_64x1024x1024_f32:idwt2_pyr_kernel:60 wavedec3_db2_L3_8x256x256x256_f32:dwt3_fwd_walk_kernel:30 waverec3_db2_L3_8x256x256x256_f32:idwt3_walk_kernel:30 wavedec2_db8_L4_64x4096x4096_f32:dwt2_fwd_stream_kernel:10 waverec2_db8_L4_64x4096x4096_f32:dwt2_inv_stream_kernel:10 fswavedec2_sym16_L5_32x8192x8192_f16:dwt2_fwd_mfma_walk_kernel:6 fswaverec2_sym16_L5_32x8192x8192_f16:idwt2_mfma_walk_kernel:6 wavedec2_db4_L3_64x1024x1024_f64:dwt2_fwd_tile_kernel:40 wavedec3_db2_L3_8x256x256x256_f64:inner_fwd_kernel:20 wavedec2_db5_L5_32x1000x1000_f32_periodic:dwt2_fwd_pyr_kernel:60 wavedec3_db5_L3_32x100x100x100_f32_periodic:dwt2_fwd_tile_kernel:40 wavedec2_bwd_db4_L3_64x1024x1024_f32:dwt2_fwd_pyr_kernel:40"}
for item in $WLS; do
  IFS=: read wl kern steps <<< "$item"
  run_pmc $wl $kern $steps
done
for item in $WLS; do
  IFS=: read wl kern steps <<< "$item"
  steps=100; case $wl in *4096x4096*|*8192x8192*) steps=20;; esac
  ( timeout 600 python bench.py --workload $wl --steps $steps --warmup 10 --no-secondary ) 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_$wl.json
  python -c "
import json
d=json.load(open('gpurun_out/${TAG}_bench_$wl.json'))
r=d['roofline']; c=d.get('cpu_baseline') or {}
print('$wl', 'ms/step', d['ms_per_step'], 'whole', d['whole_call']['frac_of_hbm_peak'], 'rotating', d['whole_call']['rotating_outputs_ms'], 'kernel frac', r['frac_unchecked'], 'traffic', r['traffic'], 'cpu', c.get('value'), c.get('cores'))"
done
