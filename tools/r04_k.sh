#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_autograd.py -q -m gpu -k "adjoint or fused or gradients_vs_reference" 2>&1 | tail -3
timeout 300 python -W ignore bench.py --workload wavedec2_bwd_db4_L3_64x1024x1024_f32 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04k_bwd.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r04k_bwd.json')); print(d['ms_per_step'], d['whole_call']['level_kernel_ms'])"
timeout 300 python -W ignore bench.py --workload waverec2_db8_L4_64x4096x4096_f32 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04k_c4rec.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r04k_c4rec.json')); print(d['ms_per_step'], d['whole_call']['level_kernel_ms'])"
