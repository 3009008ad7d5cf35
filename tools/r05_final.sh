#!/bin/bash
# end-of-round evidence with the final library: config 2 PMC + kernel stats + bench lines, the default bench run, the batch sweep
export TMPDIR=/tmp
WLS="wavedec2_db4_L3_64x1024x1024_f32:dwt2_fwd_pyr_kernel:60 waverec2_db4_L3_64x1024x1024_f32:idwt2_pyr_kernel:60 wavedec3_db2_L3_8x256x256x256_f64:dwt2_fwd_tile_kernel:20 wavedec2_bwd_db4_L3_64x1024x1024_f32:dwt2_fwd_pyr_kernel:40" TAG=r05z bash tools/r05_evidence.sh > gpurun_out/r05z_evidence.log 2>&1
grep -v "^  File\|^Traceback\|^    \|json.decoder" gpurun_out/r05z_evidence.log | tail -12
timeout 900 python bench.py > gpurun_out/r05z_bench_default_run.json 2> gpurun_out/r05z_bench_default_err.txt
O=gpurun_out/r05_batch_sweep.txt; : > $O
echo "wavedec2 db4 level 3 on B x 1024 x 1024 f32, results dropped (every call rewrites one output block):" >> $O
timeout 600 python -W ignore tools/pyr_batch_sweep.py 16,32,48,64,65,72,80,96,100,128,192,256 2>&1 | grep -v amdgpu | tee -a $O
