#!/bin/bash
export TMPDIR=/tmp
timeout 300 python -W ignore tools/pyr_compact_check.py 2>&1 | grep -v amdgpu | tee gpurun_out/r04i_compact.txt
for rep in 1 2; do for dbg in 0 2048; do
  timeout 200 python -W ignore tools/pyr_ab.py $dbg 2>&1 | grep -v amdgpu | tail -1 | tee -a gpurun_out/r04i_compact.txt
done; done
timeout 300 python -m pytest tests/test_gpu_pyramid.py tests/test_gpu_autograd.py -q -m gpu -k "small_planes_reconstruction or adjoint or fused" 2>&1 | tail -3
for lib in libmifwt.so libmifwt_border4.so; do
MIFWT_LIB=$lib timeout 300 python -W ignore bench.py --workload wavedec2_bwd_db4_L3_64x1024x1024_f32 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04i_bwd_$lib.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r04i_bwd_$lib.json')); print('$lib', d['ms_per_step'], d['whole_call']['level_kernel_ms'])"
done
