#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_pyramid.py tests/test_gpu_ipyr.py -x -q -p no:cacheprovider 2>&1 | tail -12 )
O=gpurun_out/r03e_seg0.txt; : > $O
for d in 0 128 0 128; do timeout 200 python tools/pyr_time.py db4 3 64x1024x1024 0 $d >> $O 2>&1; done
timeout 200 python tools/inv2d_time.py >> $O 2>&1
grep -v amdgpu $O
