#!/bin/bash
export TMPDIR=/tmp
TAG=r04z2 WL=waverec3_db2_L3_8x256x256x256_f32 KERNEL=idwt3_walk_kernel STEPS=30 bash tools/pmc_workload.sh 2>&1 | tail -32
