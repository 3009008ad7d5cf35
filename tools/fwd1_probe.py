"""One analysis level of a big plane: the per-level kernels (ids 7 / 1, what mifwt_dwt_fwd and the synthesis adjoint pick) against the
streaming multi-level kernel run with ONE level (id 16 through mifwt_dwt2_fwd_pyramid, MIFWT_OPT_PYRAMID_MODE 1)."""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
E = _engine.ENGINE
def t(fn, n=60):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    r = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); r.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(r)[2]
for B, N, wav, mode in ((64, 1024, 'db4', 'zero'), (64, 1024, 'db4', 'reflect'), (64, 1024, 'db2', 'zero'), (64, 515, 'db4', 'zero'), (16, 1400, 'db3', 'symmetric'), (64, 2048, 'db4', 'zero')):
    lo, hi = ptwt_amd._wavelets.host_taps(wav)[:2]
    mid = _engine.MODE_IDS[mode]
    xs = [torch.randn(B, N, N, device='cuda') for _ in range(3)]
    i = [0]
    def per_level():
        i[0] += 1; return E.analysis(xs[i[0] % 3], lo, hi, mid)
    def pyr1():
        i[0] += 1; return E.analysis_pyramid(xs[i[0] % 3], lo, hi, mid, 1)
    a = t(per_level)
    _engine.set_option(_engine.OPT_PYRAMID_MODE, 1)
    try:
        ok = pyr1() is not None
        b = t(pyr1) if ok else float('nan')
    finally:
        _engine.set_option(_engine.OPT_PYRAMID_MODE, 0)
    print(f"{B} x {N}^2 {wav} {mode}: per-level kernel id {_engine.kernel_id(2, torch.float32, mode, len(lo), B, (N, N))} {a:.1f} us, kernel 16 with one level {b:.1f} us", flush=True)
