#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r05e_matrix.txt; : > $O
timeout 900 python -W ignore tools/pyr_ab2.py 0,1,2,3,4,5,6,7 11 2>&1 | grep -v amdgpu | tee -a $O
timeout 300 python -W ignore tools/pyr_clock.py 64 1 2>&1 | grep -v amdgpu | tee -a $O
timeout 300 python -W ignore tools/pyr_clock.py 64 3 2>&1 | grep -v amdgpu | tee -a $O
