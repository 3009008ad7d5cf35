#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_walk3.py -x -q -k gradients 2>&1 | tail -12 | tee gpurun_out/r04x3_walk3_tests.txt
