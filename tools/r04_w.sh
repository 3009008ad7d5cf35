#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_walk2.py -x -q 2>&1 | tail -15 | tee gpurun_out/r04x1_walk2_tests.txt
timeout 300 python tools/iwalk2_time.py 2>&1 | tee gpurun_out/r04x1_iwalk2_time.txt
