"""Config 4's first level (64 x 4096^2 db8, kernel id 1) and its synthesis (id 2) through the C ABI with two layouts of the level
buffer: four PLANES ([B, 4, M, M]: the library's) against the four bands of a row SIDE BY SIDE ([B, M, 4, M]).  Only strides differ.
ms per launch."""
import ctypes, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptwt_amd
from ptwt_amd import _engine as E
lib = E.load_library()
B, N, wav = 64, 4096, sys.argv[1] if len(sys.argv) > 1 else 'db8'
taps = ptwt_amd._wavelets.host_taps(wav)
L = len(taps[0]); M = (N + L - 1) // 2
arr = lambda t: (ctypes.c_double * L)(*t)
dlo, dhi, rlo, rhi = (arr(t) for t in taps)
xs = [torch.randn(B, N, N, device='cuda') for _ in range(2)]
PITCH = int(sys.argv[2]) if len(sys.argv) > 2 else 0  # row pitch of the coefficient planes in samples (0: dense)
def desc(inter):
    d = E.LevelDesc()
    d.ndim, d.dtype, d.mode, d.filt_len, d.batch = 2, 0, E.MODE_IDS['reflect'], L, B
    d.sig_extent[0] = d.sig_extent[1] = N
    d.coef_extent[0] = d.coef_extent[1] = M
    d.sig_stride[0], d.sig_stride[1], d.sig_stride[2] = N * N, N, 1
    P = PITCH or M
    for st in (d.approx_stride, d.detail_stride):
        st[0], st[1], st[2] = 4 * M * P, (4 * P if inter else P), 1
    return d
def run(inter, reps=12):
    d = desc(inter)
    P = PITCH or M
    bufs = [torch.empty(B, 4 * M * P, device='cuda') for _ in range(2)]
    step = 4 * P if inter else 4 * M * P
    kid_f, kid_i = lib.mifwt_kernel_id(ctypes.byref(d), 0), lib.mifwt_kernel_id(ctypes.byref(d), 1)
    wsb = max(lib.mifwt_workspace_bytes(ctypes.byref(d), 0), lib.mifwt_workspace_bytes(ctypes.byref(d), 1))
    ws = torch.empty(max(1, wsb), dtype=torch.uint8, device='cuda')
    stream = torch.cuda.current_stream().cuda_stream
    def fwd(i):
        b = bufs[i & 1].data_ptr()
        det = (ctypes.c_void_p * 3)(b + step, b + 2 * step, b + 3 * step)
        assert lib.mifwt_dwt_fwd(ctypes.byref(d), xs[i & 1].data_ptr(), b, det, dlo, dhi, ws.data_ptr(), wsb, stream) == 0
    def inv(i):
        b = bufs[i & 1].data_ptr()
        det = (ctypes.c_void_p * 3)(b + step, b + 2 * step, b + 3 * step)
        assert lib.mifwt_dwt_inv(ctypes.byref(d), b, det, xs[i & 1].data_ptr(), rlo, rhi, ws.data_ptr(), wsb, stream) == 0
    out = []
    for f in (fwd, inv):
        for i in range(3): f(i)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(reps): f(i)
        torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / reps * 1e3)
    return kid_f, kid_i, out
for rep in range(2):
    for inter in (False, True):
        kf, ki, (tf, ti) = run(inter)
        print('pitch %d ' % (PITCH or M) + '%s: analysis (id %d) %.3f ms, synthesis (id %d) %.3f ms' % ('rows side by side' if inter else 'planes           ', kf, tf, ki, ti))
