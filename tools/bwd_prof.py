"""Where wavedec2 forward + backward w.r.t. the data (config 2, reflect) spends its time: torch profiler kernel totals."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptwt_amd
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda:0')
x = torch.randn(64, 1024, 1024, device=dev, requires_grad=True)
with torch.no_grad():
    gouts = [torch.randn_like(t) for t in [ptwt_amd.wavedec2(x, 'db4', level=3)[0]] + [t for lv in ptwt_amd.wavedec2(x, 'db4', level=3)[1:] for t in lv]]
def step():
    c = ptwt_amd.wavedec2(x, 'db4', mode='reflect', level=3)
    flat = [c[0]] + [t for lv in c[1:] for t in lv]
    return torch.autograd.grad(flat, x, gouts)
for _ in range(5): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(10): step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=16, max_name_column_width=80))
