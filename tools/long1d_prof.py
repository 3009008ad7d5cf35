"""Cycle stamps of the chunked 1-D kernel per workgroup: start, after the range bookkeeping, after the chunk is parked, after each
level, end.  usage: long1d_prof.py [wavelet] [BxN] [threads]"""
import ctypes, sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
wav = sys.argv[1] if len(sys.argv) > 1 else 'db5'
shape = tuple(int(v) for v in sys.argv[2].split('x')) if len(sys.argv) > 2 else (32, 1000000)
lev = 10 if shape[1] > 100000 else 4
if len(sys.argv) > 3: _engine.set_option(2, int(sys.argv[3]))
if len(sys.argv) > 4: _engine.set_option(11, int(sys.argv[4]))
lib = _engine.load_library()
lib.mifwt_pyr_profile_buffer.argtypes = [ctypes.c_void_p]
x = torch.randn(*shape, device='cuda')
taps = ptwt_amd._wavelets.host_taps(wav)
for _ in range(5): _engine.ENGINE.analysis_tail(x, taps[0], taps[1], _engine.MODE_IDS['periodic'], lev)
nwg = 8192
buf = torch.zeros(2 * nwg * 12, dtype=torch.int64, device='cuda')
lib.mifwt_pyr_profile_buffer(buf.data_ptr())
_engine.ENGINE.analysis_tail(x, taps[0], taps[1], _engine.MODE_IDS['periodic'], lev)
torch.cuda.synchronize()
lib.mifwt_pyr_profile_buffer(None)
b = buf[: nwg * 12].view(nwg, 12).cpu().double()
raw = buf.cpu()
used = b[:, 0] > 0
b = b[used]
print('workgroups', int(used.sum()), ' span of the launch (cycles)', float(b.max() - b[:, 0].min()))
names = ['park chunk'] + [f'level {i}' for i in range(1, 10)]
for part, sel in (('end-piece workgroups', slice(0, shape[0])), ('interior chunks', slice(shape[0], None))):
    bb = b[sel]
    n = int((bb[0] > 0).sum())
    d = bb[:, 1:n] - bb[:, : n - 1]
    print(part, ': total mean %.0f cycles' % (bb[:, n - 1] - bb[:, 0]).mean(), ' | '.join(f'{(names[i] if i < n - 2 else "approx store")} {d[:, i].mean():.0f}' for i in range(n - 1)))

nw = int(used.sum()) // 2 if (raw[nwg * 12 - 12:] != 0).any() or True else int(used.sum())
g = raw.view(-1, 12).double()
nz = (g[:, 0] > 0).nonzero().flatten()
nw = int(nz.numel())
# the fine stamps of workgroup w sit in row (grid + w); grid = number of workgroups = first gap
rows_used = nz.tolist()
grid = nw - shape[0]
if grid > shape[0]:
    f = g[grid: grid + shape[0]]
    n = int((f[0] > 0).sum())
    d = f[:, 1:n] - f[:, : n - 1]
    print('end-piece workgroups, level 2, stamps (setup / walk / end-of-row outputs per piece, then barrier):', ' | '.join('%.0f' % v for v in d.mean(0).tolist()))
