#!/bin/bash
export TMPDIR=/tmp
SCRIPT=tools/pyr_once.py KERNEL=dwt2_fwd_pyr bash tools/pmc_any.sh 2>&1 | tail -12 | tee gpurun_out/r05n_pmc_tail12.txt
