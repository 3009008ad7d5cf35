#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r05b_segs.txt; : > $O
timeout 600 python -W ignore tools/pyr_segs.py db4 64 2>&1 | grep -v amdgpu | tee -a $O
O=gpurun_out/r05b_clock.txt; : > $O
timeout 300 python -W ignore tools/pyr_clock.py 64 0 2>&1 | grep -v amdgpu | tee -a $O
timeout 300 python -W ignore tools/pyr_clock.py 64 8192 2>&1 | grep -v amdgpu | tee -a $O
timeout 1200 python -m pytest tests/test_gpu_pyramid.py -q -m gpu -x -k "pyramid" 2>&1 | tail -5 | tee gpurun_out/r05b_tests.txt
