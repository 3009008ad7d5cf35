"""Where a learnable-wavelet step's time goes (torch profiler, kernel totals), config 2's batch."""
import json, os, sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from torch.profiler import profile, ProfilerActivity
banks = json.load(open(os.path.join('tests', 'golden', 'pywt_filter_banks.json')))
dev = torch.device('cuda:0')
x = torch.randn(64, 1024, 1024, device=dev)
taps = [torch.tensor(banks['db4'][f], dtype=torch.float64, device=dev, requires_grad=True) for f in ('dec_lo', 'dec_hi', 'rec_lo', 'rec_hi')]
def step():
    xx = x.detach().requires_grad_(True)
    c = ptwt_amd.wavedec2(xx, tuple(taps), mode='reflect', level=3)
    loss = c[0].square().mean() + sum(t.square().mean() for lv in c[1:] for t in lv)
    torch.autograd.grad(loss, [xx] + taps[:2])
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=18, max_name_column_width=70))
