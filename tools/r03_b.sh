#!/bin/bash
# round 3, call 2: output layout A/B (planes vs band-interleaved rows), XCD-aware block map, compute sensitivity of the whole kernel
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r03b_layout_xcd.txt
: > $O
timeout 300 python tools/pyr_direct.py planes rows rowsP planes >> $O 2>&1
MIFWT_DBG=32 timeout 300 python tools/pyr_direct.py planes rows >> $O 2>&1
MIFWT_DBG=8 timeout 300 python tools/pyr_direct.py planes >> $O 2>&1
MIFWT_DBG=64 timeout 300 python tools/pyr_direct.py planes >> $O 2>&1
MIFWT_DBG=4 timeout 300 python tools/pyr_direct.py planes >> $O 2>&1
MIFWT_DBG=68 timeout 300 python tools/pyr_direct.py planes >> $O 2>&1
for nb in 3 5 6; do timeout 200 python tools/pyr_time.py db4 3 64x1024x1024 0 0 0 $nb >> $O 2>&1; done
grep -v amdgpu.ids $O
