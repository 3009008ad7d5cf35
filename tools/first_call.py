"""Host timeline of ONE ptwt_amd.wavedec2 call on an idle GPU (what a short timed loop pays once): call start -> the C call that
launches the kernel -> its return -> the call's return; and the same in steady state (queue busy)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptwt_amd
from ptwt_amd import _engine
dev = torch.device('cuda:0')
xs = [torch.randn(64, 1024, 1024, device=dev) for _ in range(3)]
f = lambda i: ptwt_amd.wavedec2(xs[i % 3], 'db4', mode='reflect', level=3)
for i in range(300): f(i)
torch.cuda.synchronize()
lib = _engine.load_library()
orig = lib.mifwt_dwt2_fwd_pyramid
marks = []
class W:
    def __call__(self, *a):
        t1 = time.perf_counter(); r = orig(*a); t2 = time.perf_counter(); marks.append((t1, t2)); return r
_engine._lib.mifwt_dwt2_fwd_pyramid = W()
import gc; gc.disable()
def one(idle):
    if idle: torch.cuda.synchronize()
    marks.clear(); t0 = time.perf_counter(); f(0); t3 = time.perf_counter()
    (t1, t2), = marks
    return (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6
for idle in (True, False):
    rs = []
    for _ in range(30):
        if not idle:
            for i in range(3): f(i)
        rs.append(one(idle))
    rs.sort(key=lambda r: sum(r)); m = rs[len(rs) // 2]
    print(("idle GPU   " if idle else "busy queue ") + "python before the C call %.1f us, C call (plan + hipLaunchKernel) %.1f us, python after %.1f us, total %.1f us" % (m[0], m[1], m[2], sum(m)))
