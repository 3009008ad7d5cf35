#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_walk3.py -x -q 2>&1 | tail -8 | tee gpurun_out/r04x7_tests.txt
timeout 300 python tools/walk3_run16.py 2>&1 | grep -v amdgpu | tee gpurun_out/r04x7_walk3_run16.txt
