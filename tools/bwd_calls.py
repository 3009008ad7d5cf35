"""Forward + backward (w.r.t. the data) of wavedec2 db4 level 3 on 64 x 1024^2, reflect: 100 steps after a spin-up (for a kernel trace:
tools/ktrace.sh), and the host time of a step when nothing waits for the GPU (one image)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptwt_amd
def flat(c): return [c[0]] + [t for lv in c[1:] for t in lv]
def make(b):
    x = torch.randn(b, 1024, 1024, device='cuda', requires_grad=True)
    with torch.no_grad(): g = [torch.randn_like(t) for t in flat(ptwt_amd.wavedec2(x, 'db4', mode='reflect', level=3))]
    return x, g
def step(x, g): return torch.autograd.grad(flat(ptwt_amd.wavedec2(x, 'db4', mode='reflect', level=3)), x, g)
x1, g1 = make(1)
for _ in range(200): step(x1, g1)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(1000): step(x1, g1)
t1 = time.perf_counter(); torch.cuda.synchronize()
print('host time of a forward + backward step (one image): %.1f us' % ((t1 - t0) / 1000 * 1e6))
x, g = make(64)
for _ in range(150): step(x, g)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100): step(x, g)
torch.cuda.synchronize(); print('64 images: %.1f us a step' % ((time.perf_counter() - t0) / 100 * 1e6))
