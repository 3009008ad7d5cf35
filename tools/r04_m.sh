#!/bin/bash
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -8 ) 2>&1 | tee gpurun_out/r04m_full_gpu_suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
