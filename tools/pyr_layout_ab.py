"""Kernel 16 on config 2 (64 x 1024^2 db4 level 3) through the C ABI with two layouts of the detail bands: three PLANES per level (the
library's level buffers: ten output streams per workgroup) against the three bands of an output row SIDE BY SIDE (band pointers W apart,
row stride 3 W: four streams).  Same kernel, same arithmetic; only the descriptors' strides differ.  us per launch, results dropped
(one output set) and rotating (three sets)."""
import ctypes, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptwt_amd
from ptwt_amd import _engine as E
lib = E.load_library()
B, N, L, NLEV = 64, 1024, 8, 3
lo, hi = ptwt_amd._wavelets.host_taps('db4')[:2]
lo_a, hi_a = (ctypes.c_double * L)(*lo), (ctypes.c_double * L)(*hi)
xs = [torch.randn(B, N, N, device='cuda') for _ in range(2)]
def build(interleaved, nsets):
    sets = []
    for _ in range(nsets):
        descs, bufs, rows, n = [], [], [], N
        for l in range(NLEV):
            m = (n + L - 1) // 2
            d = E.LevelDesc()
            d.ndim, d.dtype, d.mode, d.filt_len, d.batch = 2, 0, E.MODE_IDS['reflect'], L, B
            d.sig_extent[0] = d.sig_extent[1] = n
            d.coef_extent[0] = d.coef_extent[1] = m
            if l == 0:
                d.sig_stride[0], d.sig_stride[1], d.sig_stride[2] = N * N, N, 1
            else:
                d.sig_stride[0], d.sig_stride[1], d.sig_stride[2] = m_prev * m_prev, m_prev, 1  # (the approximation in between never exists)
            if interleaved:
                buf = torch.empty(B, m, 3, m, device='cuda')
                d.detail_stride[0], d.detail_stride[1], d.detail_stride[2] = 3 * m * m, 3 * m, 1
                ptrs = [buf.data_ptr() + 4 * m * b for b in range(3)]
            else:
                buf = torch.empty(B, 3, m, m, device='cuda')
                d.detail_stride[0], d.detail_stride[1], d.detail_stride[2] = 3 * m * m, m, 1
                ptrs = [buf.data_ptr() + 4 * m * m * b for b in range(3)]
            d.approx_stride[0], d.approx_stride[1], d.approx_stride[2] = m * m, m, 1
            descs.append(d); bufs.append(buf); rows.append((ctypes.c_void_p * 3)(*ptrs))
            n, m_prev = m, m
        approx = torch.empty(B, n, n, device='cuda')
        refs = (ctypes.POINTER(E.LevelDesc) * NLEV)(*[ctypes.pointer(d) for d in descs])
        det = (ctypes.POINTER(ctypes.c_void_p) * NLEV)(*[ctypes.cast(r, ctypes.POINTER(ctypes.c_void_p)) for r in rows])
        sets.append((descs, bufs, rows, approx, refs, det))
    return sets
def run(sets, reps=200):
    stream = torch.cuda.current_stream().cuda_stream
    def call(i):
        descs, bufs, rows, approx, refs, det = sets[i % len(sets)]
        rc = lib.mifwt_dwt2_fwd_pyramid(NLEV, refs, xs[i & 1].data_ptr(), det, approx.data_ptr(), lo_a, hi_a, stream)
        assert rc == 0, rc
    for i in range(50): call(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(reps): call(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
ok = lib.mifwt_dwt2_fwd_pyramid_supported(NLEV, build(True, 1)[0][4])
print('interleaved layout served by route', ok)
for rep in range(3):
    print('planes: dropped %.1f rotating %.1f    rows side by side: dropped %.1f rotating %.1f' % (
        run(build(False, 1)), run(build(False, 3)), run(build(True, 1)), run(build(True, 3))))
