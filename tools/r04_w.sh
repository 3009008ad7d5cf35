#!/bin/bash
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_gpu_walk3.py tests/test_gpu_parity.py tests/test_gpu_autograd.py tests/test_gpu_graphs.py -x -q 2>&1 | tail -8 ) 2>&1 | tee gpurun_out/r04w10_tests.txt
timeout 300 python tools/walk3_levels.py 2>&1 | head -3 | tee gpurun_out/r04w10_levels.txt
