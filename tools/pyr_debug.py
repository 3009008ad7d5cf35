import sys, numpy as np, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
from oracle import fwt_oracle as O
from tests import _golden as G
torch.manual_seed(0)
wav = sys.argv[1] if len(sys.argv) > 1 else 'haar'
lev = int(sys.argv[2]) if len(sys.argv) > 2 else 1
shape = tuple(int(v) for v in sys.argv[3].split('x')) if len(sys.argv) > 3 else (1, 64, 64)
mode = sys.argv[4] if len(sys.argv) > 4 else 'reflect'
seg = int(sys.argv[5]) if len(sys.argv) > 5 else 0
x = torch.randn(*shape)
if seg: _engine.set_option(_engine.OPT_PAIR_ROWS, seg)
_engine.level_events = []
got = ptwt_amd.wavedec2(x.cuda(), wav, mode=mode, level=lev)
torch.cuda.synchronize()
print('kids', [e[1] for e in _engine.level_events])
want = O.wavedec2(x.numpy().astype(np.float64), wav, mode=mode, level=lev)
for (n, a), (_, b) in zip(G.flatten_coeffs(got), G.flatten_coeffs(want)):
    a = a.cpu().numpy().astype(np.float64)
    err = np.abs(a - b)
    bad = err > 1e-4 * (np.abs(b).max() + 1e-30)
    print(n, a.shape, 'relerr %.3e' % G.relerr(a, b), 'bad', int(bad.sum()), 'of', bad.size)
    if bad.any():
        rows = np.unique(np.nonzero(bad)[-2]); cols = np.unique(np.nonzero(bad)[-1])
        print('   bad rows', rows[:12], '...', rows[-4:], ' bad cols', cols[:12], '...', cols[-4:])
        i = tuple(np.argwhere(bad)[0]); print('   first bad', i, 'got', a[i], 'want', b[i])
