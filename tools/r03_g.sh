#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03g_store_policy_nbuf.txt; : > $O
for lib in libmifwt.so libmifwt_st16.so libmifwt_st17.so libmifwt.so libmifwt_st17.so; do
  echo "== MIFWT_LIB=$lib" >> $O
  MIFWT_LIB=$lib timeout 200 python tools/pyr_time.py >> $O 2>&1
  MIFWT_LIB=$lib timeout 200 python tools/inv2d_time.py 2>&1 | head -2 >> $O
done
for nb in 4 5 6; do MIFWT_NBUF=$nb timeout 200 python tools/inv2d_time.py 2>&1 | head -2 >> $O; done
for d in 1 2 4; do MIFWT_DBG=$d timeout 200 python tools/inv2d_time.py 2>&1 | head -2 >> $O; done
grep -v amdgpu $O
