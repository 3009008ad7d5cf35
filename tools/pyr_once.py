"""A few config-2 wavedec2 calls (64 x 1024^2 db4 level 3): the workload under tools/gpu_pmc_pyr.sh."""
import sys, torch
sys.path.insert(0, __file__.rsplit('/', 2)[0])
import ptwt_amd
xs = [torch.randn(64, 1024, 1024, device='cuda') for _ in range(3)]
for i in range(8): ptwt_amd.wavedec2(xs[i % 3], 'db4', level=3)
torch.cuda.synchronize()
