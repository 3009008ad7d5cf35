"""One synthesis level of a big plane: the per-level kernels (ids 2 / 8, what mifwt_dwt_inv and the analysis adjoint pick) against the
streaming multi-level kernel run with ONE level (id 22 through mifwt_dwt2_inv_pyramid)."""
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
for B, N, wav in ((64, 1024, 'db4'), (64, 515, 'db4'), (64, 1024, 'db2'), (32, 1000, 'db5'), (64, 2055, 'db4'), (16, 1400, 'db3')):
    lo, hi = ptwt_amd._wavelets.host_taps(wav)[2:]
    L = len(lo); M = (N + L - 1) // 2
    bufs = [torch.randn(B, 4, M, M, device='cuda') for _ in range(3)]
    i = [0]
    def per_level():
        b = bufs[i[0] % 3]; i[0] += 1
        return E.synthesis(b[:, 0], [b[:, 1], b[:, 2], b[:, 3]], lo, hi, [N, N])
    def pyr1():
        b = bufs[i[0] % 3]; i[0] += 1
        return E.synthesis_pyramid(b[:, 0], [[b[:, 1], b[:, 2], b[:, 3]]], lo, hi, [N, N])
    y1, y2 = per_level(), pyr1()
    if y2 is None:
        print(B, N, wav, 'pyramid route not served'); continue
    kid = _engine.kernel_id(2, torch.float32, 'zero', L, B, (N, N), direction=1)
    print(f"{B} x {N}^2 {wav}: per-level kernel id {kid} {t(per_level):.1f} us, kernel 22 with one level {t(pyr1):.1f} us, max diff {float((y1 - y2).abs().max()):.2e}", flush=True)
