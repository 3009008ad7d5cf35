#!/bin/bash
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_pyramid.py -q -m gpu -x -k "pyramid" 2>&1 | tail -8 | tee gpurun_out/r05f_tests.txt
O=gpurun_out/r05f_ab.txt; : > $O
timeout 600 python -W ignore tools/pyr_ab2.py 0,8192 11 2>&1 | grep -v amdgpu | tee -a $O
O=gpurun_out/r05f_clock.txt; : > $O
timeout 300 python -W ignore tools/pyr_clock.py 64 0 2>&1 | grep -v amdgpu | tee -a $O
O=gpurun_out/r05f_batch_sweep.txt; : > $O
timeout 600 python -W ignore tools/pyr_batch_sweep.py 2>&1 | grep -v amdgpu | tee -a $O
