#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r05a_clock.txt; : > $O
timeout 300 python -W ignore tools/pyr_clock.py 64 0 2>&1 | grep -v amdgpu | tee -a $O
timeout 300 python -W ignore tools/pyr_clock.py 64 4 2>&1 | grep -v amdgpu | tee -a $O
timeout 300 python -W ignore tools/pyr_clock.py 16 0 2>&1 | grep -v amdgpu | tee -a $O
O=gpurun_out/r05a_widths.txt; : > $O
timeout 600 python -W ignore tools/pyr_widths.py 2>&1 | grep -v amdgpu | tee -a $O
O=gpurun_out/r05a_batch_sweep_before.txt; : > $O
timeout 600 python -W ignore tools/pyr_batch_sweep.py 2>&1 | grep -v amdgpu | tee -a $O
