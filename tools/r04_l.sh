#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_autograd.py -q -m gpu -k "second_order or double_backward or tap" 2>&1 | tail -12
