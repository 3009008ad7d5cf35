"""How a K = 20 timed loop of config 2 ends: torch.cuda.synchronize() (blocking wait) against polling an event recorded behind the last
call, then synchronising.  us per step."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptwt_amd
x = torch.randn(64, 1024, 1024, device='cuda')
f = lambda: ptwt_amd.wavedec2(x, 'db4', mode='reflect', level=3)
for i in range(300): f()
torch.cuda.synchronize()
import gc; gc.disable()
def loop(k, poll):
    for i in range(5): f()
    torch.cuda.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(k): f()
    if poll:
        ev = torch.cuda.Event(); ev.record()
        while not ev.query(): pass
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e6
for rep in range(3):
    a = sorted(loop(20, False) for _ in range(15)); b = sorted(loop(20, True) for _ in range(15)); c = sorted(loop(200, False) for _ in range(5))
    print('K=20 synchronize: median %.2f min %.2f   K=20 polled event + synchronize: median %.2f min %.2f   K=200: %.2f' % (a[7], a[0], b[7], b[0], c[2]))
