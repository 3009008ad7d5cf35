#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r04t_segment_balance.txt; : > $O
for rep in 1 2 3; do for dbg in 0 4096; do
  timeout 200 python -W ignore tools/pyr_ab.py $dbg 2>&1 | grep -v amdgpu | tail -1 | tee -a $O
done; done
timeout 900 python -m pytest tests/test_gpu_pyramid.py tests/test_gpu_autograd.py tests/test_gpu_parity.py -q -m gpu -k "pyramid or backward_routes or extension_loaded" 2>&1 | tail -4
