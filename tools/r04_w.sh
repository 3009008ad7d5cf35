#!/bin/bash
export TMPDIR=/tmp
for i in 1 2; do
for lib in libmifwt.so libmifwt_dmadef.so; do
  MIFWT_LIB=$lib timeout 120 python tools/pyr_ab.py 2>&1 | grep -v "Warning\|warn\|amdgpu.ids"
done
done | tee gpurun_out/r04x2_dma_policy_c2.txt
cat > /tmp/rec.py <<'PY'
import os, sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from tools.walk3_time import t
xs = [torch.randn(64, 1024, 1024, device='cuda') for _ in range(3)]
cs = [ptwt_amd.wavedec2(x, 'db4', level=3) for x in xs]
print(os.environ.get('MIFWT_LIB'), f"waverec2 config 2: {t(lambda c: ptwt_amd.waverec2(c, 'db4'), cs, 100):.1f} us", flush=True)
PY
for lib in libmifwt.so libmifwt_dmadef.so libmifwt.so libmifwt_dmadef.so; do MIFWT_LIB=$lib timeout 120 python /tmp/rec.py 2>&1 | grep waverec2; done | tee -a gpurun_out/r04x2_dma_policy_c2.txt
