#!/bin/bash
# end-of-session evidence run: parity tests, bench lines for every workload, secondary benches, PMC + kernel trace
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 300 python __graft_entry__.py --smoke ) > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log
( timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -2 gpurun_out/pytest_gpu.log
( timeout 300 python bench.py ) > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log | cut -c1-1500
for wl in wavedec3_db2_L3_8x256x256x256_f32 wavedec2_db8_L4_64x4096x4096_f32 fswavedec2_sym16_L5_32x8192x8192_f16 wavedec_db5_L10_32x1000000_f32; do
  ( timeout 600 python bench.py --steps 20 --warmup 5 --workload $wl --no-cpu-baseline ) 2>/dev/null | tail -1 > gpurun_out/bench_$wl.log
  python -c "import json,sys; d=json.loads(open('gpurun_out/bench_$wl.log').read()); print('$wl', d['ms_per_step'], d['value'], d['whole_call']['frac_of_hbm_peak'], d['roofline']['frac'])"
done
( timeout 600 python tools/bench_more.py ) 2>/dev/null > gpurun_out/bench_more.log
( timeout 120 python tools/copy_floor.py ) 2>/dev/null > gpurun_out/copy_floor.log
( timeout 300 python tools/host_overhead.py ) 2>/dev/null | head -3 > gpurun_out/host_overhead.log
PAIR=1 RPC=0 DEPTH=0 bash tools/gpu_pmc.sh > gpurun_out/pmc.log 2>&1; tail -3 gpurun_out/pmc.log
