#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r05c_ab.txt; : > $O
timeout 600 python -W ignore tools/pyr_ab2.py 0,16384,32768,49152,4,16388 2>&1 | grep -v amdgpu | tee -a $O
O=gpurun_out/r05c_clock.txt; : > $O
timeout 300 python -W ignore tools/pyr_clock.py 64 0 2>&1 | grep -v amdgpu | tee -a $O
timeout 300 python -W ignore tools/pyr_clock.py 64 49152 2>&1 | grep -v amdgpu | tee -a $O
