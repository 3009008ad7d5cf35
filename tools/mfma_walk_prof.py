"""Per-wave cycle profile of the walking matrix-core kernel on level 1 of the config-5 slice (profiling instantiation)."""
import ctypes, sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
ptwt_amd.set_half_storage(True)
dbgs = [int(v) for v in sys.argv[1:]] or [0]
lib = _engine.load_library()
lib.mifwt_pyr_profile_buffer.argtypes = [ctypes.c_void_p]
x = torch.randn(32, 8192, 8192, device='cuda').half()
for _ in range(2): ptwt_amd.wavedec2(x, 'sym16', mode='reflect', level=1)
for dbg in dbgs:
    _engine.set_option(_engine.OPT_DEBUG, dbg)
    buf = torch.zeros(1024 * 5 * 8, dtype=torch.int64, device='cuda')
    lib.mifwt_pyr_profile_buffer(buf.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ptwt_amd.wavedec2(x, 'sym16', mode='reflect', level=1); e1.record()
    torch.cuda.synchronize()
    lib.mifwt_pyr_profile_buffer(None)
    b = buf.view(1024, 5, 8).double().cpu()
    m, l = b[:, :4].mean(dim=(0, 1)) / 1e3, b[:, 4].mean(dim=0) / 1e3
    print(f"debug {dbg}: {e0.elapsed_time(e1):.3f} ms")
    print("  matrix waves (k cycles): setup %.0f  bookkeeping %.0f  barrier A %.0f  horizontal %.0f  barrier B %.0f  vertical+stores %.0f   total %.0f" % (*m[:6].tolist(), float(m[:6].sum())))
    print("  loader      (k cycles): setup %.0f  wait chunk %.0f  patch %.0f  barrier A %.0f  barrier B %.0f  requests %.0f   total %.0f" % (*l[:6].tolist(), float(l[:6].sum())))
_engine.set_option(_engine.OPT_DEBUG, 0)
