#!/bin/bash
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_pyramid.py -q -m gpu -x -k "pyramid" 2>&1 | tail -12 | tee gpurun_out/r05k_tests.txt
O=gpurun_out/r05k_ab.txt; : > $O
timeout 600 python -W ignore tools/pyr_ab2.py 0,524288,1048576 11 2>&1 | grep -v amdgpu | tee -a $O
timeout 300 python -W ignore tools/pyr_clock.py 64 0 2>&1 | grep -v amdgpu | tee gpurun_out/r05k_clock.txt
