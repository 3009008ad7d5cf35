"""Wave priorities in the streaming kernels of config 2 (MIFWT_OPT_DEBUG: 16 = the loader waves at the default priority, 32 (synthesis
only) = the synthesis waves raised): whole calls, alternating rounds."""
import sys, time, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
xs = [torch.randn(64, 1024, 1024, device='cuda') for _ in range(3)]
cs = [ptwt_amd.wavedec2(x, 'db4', level=3) for x in xs]
def loop(fn):
    for i in range(20): fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(200): fn(i)
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / 200
for rnd in range(3):
    for dbg in (0, 16, 32, 48):
        _engine.set_option(_engine.OPT_DEBUG, dbg)
        a = loop(lambda i: ptwt_amd.wavedec2(xs[i % 3], 'db4', level=3)) if dbg in (0, 16) else float('nan')
        b = loop(lambda i: ptwt_amd.waverec2(cs[i % 3], 'db4'))
        print(f"round {rnd} debug {dbg:2d}: wavedec2 {a:6.1f} us   waverec2 {b:6.1f} us", flush=True)
_engine.set_option(_engine.OPT_DEBUG, 0)
