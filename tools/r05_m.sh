#!/bin/bash
export TMPDIR=/tmp
for dbg in 0 3 1048576; do timeout 300 python -W ignore tools/pyr_clock.py 64 $dbg 2>&1 | grep -v amdgpu | grep -v "launch [0-4]" | tee -a gpurun_out/r05m_l2_sections.txt; done
