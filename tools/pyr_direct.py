"""Config-2 pyramid launch (64 x 1024^2 db4, 3 levels) straight through the C ABI with a choice of OUTPUT LAYOUTS, timed back to
back and checked against the engine's own result:
   planes   [B, nb, H, W]          (what the engine allocates: a level's bands are planes of one buffer)
   rows     [B, H, nb, W]          (the bands of one coefficient row next to one another: one write stream per level instead of three)
   rowsP    rows with the row of a band padded to a multiple of 4 floats
usage: pyr_direct.py [layout ...]   (MIFWT_DBG = value for MIFWT_OPT_DEBUG)"""
import ctypes, os, sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
from ptwt_amd._engine import LevelDesc

lib = _engine.load_library()
B, H0, W0, L, NLEV = 64, 1024, 1024, 8, 3
wav = 'db4'
mode = 'reflect'
dbg = int(os.environ.get('MIFWT_DBG', '0'))
if dbg: _engine.set_option(_engine.OPT_DEBUG, dbg)
lo, hi = ptwt_amd._wavelets.host_taps(wav)[:2]
lo_a, hi_a = (ctypes.c_double * L)(*lo), (ctypes.c_double * L)(*hi)
ext = [(H0, W0)]
for _ in range(NLEV): ext.append(((ext[-1][0] + L - 1) // 2, (ext[-1][1] + L - 1) // 2))


def build(layout, x):
    descs, bufs, rows = [], [], []
    for l in range(NLEV):
        Hs, Ws = ext[l]
        Hc, Wc = ext[l + 1]
        nb = 4 if l == NLEV - 1 else 3
        pw = (Wc + 3) & ~3 if layout == 'rowsP' else Wc
        if layout == 'planes':
            buf = torch.empty(B, nb, Hc, Wc, device='cuda')
            sb, sh, band = nb * Hc * Wc, Wc, Hc * Wc
            view = buf
        else:
            buf = torch.empty(B, Hc, nb, pw, device='cuda')
            sb, sh, band = Hc * nb * pw, nb * pw, pw
            view = buf.permute(0, 2, 1, 3)[..., :Wc]
        d = LevelDesc()
        d.ndim, d.dtype, d.mode, d.filt_len, d.batch = 2, 0, _engine.MODE_IDS[mode], L, B
        d.sig_extent[0], d.sig_extent[1] = Hs, Ws
        d.coef_extent[0], d.coef_extent[1] = Hc, Wc
        d.sig_stride[0], d.sig_stride[1], d.sig_stride[2] = (x.stride(0), x.stride(1), 1) if l == 0 else (0, Ws, 1)
        for s in (d.approx_stride, d.detail_stride):
            s[0], s[1], s[2] = sb, sh, 1
        descs.append(d)
        bufs.append((buf, view, nb))
        base = buf.data_ptr() + (nb - 3) * band * 4
        rows.append((ctypes.c_void_p * 3)(base, base + band * 4, base + 2 * band * 4))
    refs = (ctypes.POINTER(LevelDesc) * NLEV)(*[ctypes.pointer(d) for d in descs])
    det = (ctypes.POINTER(ctypes.c_void_p) * NLEV)(*[ctypes.cast(r, ctypes.POINTER(ctypes.c_void_p)) for r in rows])
    return descs, refs, det, rows, bufs


def run(layout):
    xs = [torch.randn(B, H0, W0, device='cuda') for _ in range(3)]
    sets = [build(layout, x) for x in xs]
    stream = torch._C._cuda_getCurrentRawStream(0)
    route = lib.mifwt_dwt2_fwd_pyramid_supported(NLEV, sets[0][1])
    def launch(i):
        descs, refs, det, rows, bufs = sets[i % 3]
        rc = lib.mifwt_dwt2_fwd_pyramid(NLEV, refs, xs[i % 3].data_ptr(), det, bufs[-1][0].data_ptr(), lo_a, hi_a, stream)
        assert rc == 0, rc
    for i in range(30): launch(i)
    torch.cuda.synchronize()
    res = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(100): launch(i)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 100 * 1e3)
    res.sort()
    # parity against the engine's own call (same kernel, engine layout) unless a debug switch changes the arithmetic / drops traffic
    err = -1.0
    if not (dbg & (1 | 2 | 4 | 8 | 64)):
        launch(0); torch.cuda.synchronize()
        want = ptwt_amd.wavedec2(xs[0], wav, level=NLEV, mode=mode)
        descs, refs, det, rows, bufs = sets[0]
        err = 0.0
        err = max(err, float((bufs[-1][1][:, 0] - want[0]).abs().max()))
        for l in range(NLEV):
            buf, view, nb = bufs[l]
            for b in range(3):
                err = max(err, float((view[:, nb - 3 + b] - want[NLEV - l][b]).abs().max()))
    byts = 4 * B * (H0 * W0 + sum(3 * h * w for h, w in ext[1:]) + ext[-1][0] * ext[-1][1])
    print(f"layout {layout:7s} dbg={dbg} route={route}: median {res[3]:.1f} us  min {res[0]:.1f} us -> {byts/res[3]/8e6:.3f} of 8 TB/s   max|diff| vs engine {err:.2e}")


for lay in (sys.argv[1:] or ['planes', 'rows', 'rowsP']):
    run(lay)
