#!/bin/bash
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -4 ) 2>&1 | tee gpurun_out/r04_final_gpu_suite.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a gpurun_out/r04_final_gpu_suite.txt
