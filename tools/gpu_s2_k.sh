#!/bin/bash
export TMPDIR=/tmp
run() { ( timeout 300 python tools/level_bench.py "$@" --rounds 3 --iters 10 ) 2>/dev/null | cut -c1-175; }
run --shape 64,4096,4096 --wavelet db4 --tile 2,1 --tr 12,16
run --shape 64,4096,4096 --wavelet db5 --tile 2,1 --tr 12,16,24
run --shape 64,4096,4096 --wavelet db6 --tile 2,1 --tr 12,16,24
run --shape 64,4096,4096 --wavelet db8 --tile 1 --tr 20,24
run --shape 64,1024,1024 --wavelet db5 --tile 2,1 --tr 12,16
run --shape 64,1024,1024 --wavelet db6 --tile 2,1 --tr 12,16
run --shape 64,1024,1024 --wavelet db7 --tile 2,1 --tr 12,16
run --shape 64,1024,1024 --wavelet db8 --tile 2,1 --tr 12,16
run --shape 64,2055,2055 --wavelet db6 --tile 2,1 --tr 12,16
