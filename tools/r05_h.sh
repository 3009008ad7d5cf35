#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r05h_tails_off.txt; : > $O
timeout 900 python -W ignore tools/pyr_ab2.py 0,65536,131072,262144,393216,458752 11 2>&1 | grep -v amdgpu | tee -a $O
