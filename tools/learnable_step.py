"""A learnable-wavelet training step on config 2's batch (64 x 1024^2 f32, db4 level 3, reflect): wavedec2 forward + backward w.r.t. the
data AND the four filter tensors (nn.Parameter-style leaves on the GPU), with the taps read by the kernels from device memory
(set_device_taps "auto": no host synchronisation) against the same step with the taps read back to the host per call ("never").
Prints ms per step and the kernel ids of one step."""
import json, os, sys, time, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
banks = json.load(open(os.path.join('tests', 'golden', 'pywt_filter_banks.json')))
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x = torch.randn(B, 1024, 1024, device=dev)
def step(taps, rec):
    xx = x.detach().requires_grad_(True)
    c = ptwt_amd.wavedec2(xx, tuple(taps), mode='reflect', level=3)
    if rec:
        loss = ptwt_amd.waverec2(c, tuple(taps)).square().mean()
    else:
        loss = c[0].square().mean() + sum(t.square().mean() for lv in c[1:] for t in lv)
    return torch.autograd.grad(loss, [xx] + (taps if rec else taps[:2]))
for rec in (False, True):
    for mode in ('auto', 'never', 'auto', 'never'):
        ptwt_amd.set_device_taps(mode)
        taps = [torch.tensor(banks['db4'][f], dtype=torch.float64, device=dev, requires_grad=True) for f in ('dec_lo', 'dec_hi', 'rec_lo', 'rec_hi')]
        for _ in range(5): step(taps, rec)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 30
        for _ in range(n): step(taps, rec)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / n * 1e3
        _engine.level_events = []
        step(taps, rec)
        kids = sorted({e[1] for e in _engine.level_events}); _engine.level_events = None
        print(f"{'wavedec2 + waverec2' if rec else 'wavedec2'} forward + backward, taps {'on the device' if mode == 'auto' else 'read to the host'}: {ms:.3f} ms per step, kernel ids {kids}")
ptwt_amd.set_device_taps('auto')
