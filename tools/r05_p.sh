#!/bin/bash
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_autograd.py -q -m gpu -x 2>&1 | tail -30 | tee gpurun_out/r05p_autograd.txt
