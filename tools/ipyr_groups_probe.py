"""What would COLUMN GROUPS cost the streaming synthesis kernel in DRAM locality?  The same launch on (a) 64 dense 1024 x 1024 planes and
(b) the four 1024-column quarters of 16 planes of 4096 columns (views: row pitch 4096 floats for the output, 2055 / 1035 / 525 for the
coefficient planes), through the C ABI.  Not a full column-group transform (the quarters ignore their halos), only the traffic pattern."""
import ctypes, sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
from ptwt_amd._engine import LevelDesc
lib = _engine.load_library()
L, NLEV = 8, 3
lo, hi = ptwt_amd._wavelets.host_taps('db4')[2:]
lo_a, hi_a = (ctypes.c_double * L)(*lo), (ctypes.c_double * L)(*hi)
def exts(H, W):
    e = [(H, W)]
    for _ in range(NLEV): e.append(((e[-1][0] + L - 1) // 2, (e[-1][1] + L - 1) // 2))
    return e
def make_call(B, H, W, wide):
    """coefficient planes of a (H, W) transform; wide = factor by which the allocated planes are wider than what the launch touches"""
    e = exts(H, W)
    descs, keep, rows = [], [], []
    approx = None
    for lvl in range(NLEV, 0, -1):  # coarsest first
        Mh, Mw = e[lvl]
        Oh, Ow = e[lvl - 1]
        pitch = (exts(H, W * wide)[lvl][1]) if wide > 1 else Mw
        buf = torch.randn(B, 4, Mh, pitch, device='cuda')
        keep.append(buf)
        d = LevelDesc()
        d.ndim, d.dtype, d.mode, d.filt_len, d.batch = 2, 0, 0, L, B
        d.coef_extent[0], d.coef_extent[1] = Mh, Mw
        d.sig_extent[0], d.sig_extent[1] = Oh, Ow
        for st in (d.approx_stride, d.detail_stride):
            st[0], st[1], st[2] = 4 * Mh * pitch, pitch, 1
        d.sig_stride[0], d.sig_stride[1], d.sig_stride[2] = H * W * wide, W * wide, 1
        descs.append(d)
        base = buf.data_ptr()
        pb = Mh * pitch * 4
        rows.append((ctypes.c_void_p * 3)(base + pb, base + 2 * pb, base + 3 * pb))
        if lvl == NLEV: approx = base
    y = torch.empty(B, H, W * wide, device='cuda')
    refs = (ctypes.POINTER(LevelDesc) * NLEV)(*[ctypes.pointer(d) for d in descs])
    det = (ctypes.POINTER(ctypes.c_void_p) * NLEV)(*[ctypes.cast(r, ctypes.POINTER(ctypes.c_void_p)) for r in rows])
    return dict(descs=descs, refs=refs, det=det, rows=rows, keep=keep, y=y, approx=approx, W=W, wide=wide)
def launch(c, g=0):
    stream = torch._C._cuda_getCurrentRawStream(0)
    off = 4 * c['W'] * g  # byte offset of the column group in the output; coefficient planes: the group's first column
    dets = []
    e = None
    rc = lib.mifwt_dwt2_inv_pyramid(NLEV, c['refs'], c['approx'], c['det'], c['y'].data_ptr() + off, lo_a, hi_a, stream)
    assert rc == 0, rc
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    res = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(res)[2]
_engine.set_option(_engine.OPT_PYRAMID_MODE, 1)
dense = [make_call(64, 1024, 1024, 1) for _ in range(3)]
print('route', lib.mifwt_dwt2_inv_pyramid_supported(NLEV, dense[0]['refs']))
i = [0]
def f_dense():
    i[0] += 1; launch(dense[i[0] % 3])
print(f"64 dense planes of 1024 x 1024: {t(f_dense):.1f} us per launch")
del dense; torch.cuda.empty_cache()
wide = [make_call(16, 1024, 1024, 4) for _ in range(3)]
print('route', lib.mifwt_dwt2_inv_pyramid_supported(NLEV, wide[0]['refs']))
def f_wide():
    i[0] += 1
    for g in range(4): launch(wide[i[0] % 3], g)
print(f"16 planes of 1024 x 4096, four launches of one 1024-column quarter each (output pitch 4096, coefficient pitches of a 4096-wide transform): {t(f_wide):.1f} us for the four")
