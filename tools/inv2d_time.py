"""waverec2 on config 2's coefficients: whole call and per launch (event pair around every launch)."""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
wav = sys.argv[1] if len(sys.argv) > 1 else 'db4'
lev = int(sys.argv[2]) if len(sys.argv) > 2 else 3
shape = tuple(int(v) for v in sys.argv[3].split('x')) if len(sys.argv) > 3 else (64, 1024, 1024)
import os
if os.environ.get('MIFWT_NBUF'): _engine.set_option(2, int(os.environ['MIFWT_NBUF']))
if os.environ.get('MIFWT_DBG'): _engine.set_option(11, int(os.environ['MIFWT_DBG']))
cs = [ptwt_amd.wavedec2(torch.randn(*shape, device='cuda'), wav, level=lev) for _ in range(3)]
for i in range(10): ptwt_amd.waverec2(cs[i % 3], wav)
torch.cuda.synchronize()
res = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(50): ptwt_amd.waverec2(cs[i % 3], wav)
    e1.record(); torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / 50 * 1e3)
res.sort()
print(f"waverec2 {wav} L{lev} {shape} lib={os.environ.get('MIFWT_LIB','-')} nbuf={os.environ.get('MIFWT_NBUF','-')} dbg={os.environ.get('MIFWT_DBG','-')}: median {res[2]:.1f} us min {res[0]:.1f} us")
_engine.level_events = []
for i in range(20): ptwt_amd.waverec2(cs[i % 3], wav)
torch.cuda.synchronize()
ev, _engine.level_events = _engine.level_events, None
agg = {}
for tag, kid, ext, s, e in ev:
    agg.setdefault((kid, tuple(ext)), []).append(s.elapsed_time(e) * 1e3)
for k, v in agg.items():
    print('  kernel id %d, extent %s: %.1f us' % (k[0], k[1], sorted(v)[len(v) // 2]))
