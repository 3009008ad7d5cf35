"""Per-wave cycle profile of one mifwt_dwt2_fwd_pyramid launch on config 2: where do the level-1 / deep waves spend their time?"""
import ctypes, sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
dbg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
lib = _engine.load_library()
lib.mifwt_pyr_profile_buffer.argtypes = [ctypes.c_void_p]
x = torch.randn(64, 1024, 1024, device='cuda')
for _ in range(5): ptwt_amd.wavedec2(x, 'db4', level=3)
nwg, nwave = 256, 16
buf = torch.zeros(nwg * nwave * 2, dtype=torch.int64, device='cuda')
if dbg: _engine.set_option(11, dbg)
lib.mifwt_pyr_profile_buffer(buf.data_ptr())
ptwt_amd.wavedec2(x, 'db4', level=3)
torch.cuda.synchronize()
lib.mifwt_pyr_profile_buffer(None)
b = buf.view(nwg, nwave, 2).cpu().double()
roles = ['L1'] * 5 + ['L2'] * 3 + ['--'] + ['L3'] * 3 + ['--', 'L1', 'ld', 'ld']  # wave -> role (mifwt_dwt2_fwd_pyr.hip pyr_role)
print('dbg', dbg)
for seg in range(4):
    sel = b[seg::4]
    tot, wait = sel[..., 0], sel[..., 1]
    print(f' segment {seg}: ' + '  '.join(f'w{w}{roles[w]}: {tot[:, w].mean()/1e3:.0f}k/{100*wait[:, w].mean()/max(tot[:, w].mean(),1):.0f}%' for w in range(12) if roles[w] != '--'))
print(' all WGs: L1 total cycles mean %.0f max %.0f ; barrier share L1 %.1f%% deep %.1f%%' % (
    b[:, :5, 0].mean(), b[:, :5, 0].max(), 100 * b[:, :5, 1].sum() / b[:, :5, 0].sum(),
    100 * b[:, [5, 6, 7, 9, 10, 11], 1].sum() / b[:, [5, 6, 7, 9, 10, 11], 0].sum()))
