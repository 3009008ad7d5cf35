#!/bin/bash
# round-6 evidence: (1) PMC + kernel-trace summary of the headline workload (tools/pmc_workload.sh), (2) kernel traces of the headline
# call in ONE output regime each (dropped / rotating: tools/trace_regime.py) whose averages are the statistic, (3) default bench lines.
# Everything under gpurun_out/ (TAG_*); every profiler run under its own timeout.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${TAG:-r06g}
WL=wavedec2_db4_L3_64x1024x1024_f32
mkdir -p $R/gpurun_out
cd $R
for item in ${WLS:-"$WL:dwt2_fwd_pyr_kernel:60 waverec2_db4_L3_64x1024x1024_f32:idwt2_pyr_kernel:60"}; do
  IFS=: read wl kern steps <<< "$item"
  TAG=$TAG WL=$wl KERNEL=$kern STEPS=$steps timeout 1500 bash tools/pmc_workload.sh > gpurun_out/${TAG}_pmc_$wl.log 2>&1
  python -c "import json; d=json.load(open('gpurun_out/${TAG}_pmc_$wl.json')); print('$wl', 'traffic', d.get('hbm_traffic_bytes'), d.get('launch_ns_by_grid'))"
done
for regime in dropped rotating; do
  O=/tmp/tr_$regime; rm -rf $O
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -o tr -- python $R/tools/trace_regime.py $regime 400 ) > gpurun_out/${TAG}_trace_$regime.log 2>&1
  f=$(find $O -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python tools/trace_regime_summary.py $f gpurun_out/${TAG}_kernel_stats_${WL}_$regime.csv $regime
done
for i in 0 1 2; do python bench.py > gpurun_out/${TAG}_bench_default_$i.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/${TAG}_bench_default_$i.json').read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], d['whole_call']['rotating_outputs_ms'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline']['traffic'])"; done
