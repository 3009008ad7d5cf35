"""Wall-clock anatomy of consecutive mifwt_dwt2_fwd_pyramid launches on config 2 (profiling instance of kernel 16): when does every
workgroup start and end (100 MHz s_memrealtime), where did it run, how long is the gap between two launches."""
import ctypes, sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dbg = int(sys.argv[2]) if len(sys.argv) > 2 else 0
lib = _engine.load_library()
lib.mifwt_pyr_profile_buffer.argtypes = [ctypes.c_void_p]
xs = [torch.randn(B, 1024, 1024, device='cuda') for _ in range(3)]
if dbg: _engine.set_option(11, dbg)
ex = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if ex: _engine.set_option(15, ex)
for i in range(10): ptwt_amd.wavedec2(xs[i % 3], 'db4', level=3)
torch.cuda.synchronize()
nl = 6
nwg = 320
bufs = [torch.zeros(nwg * 16 * 2, dtype=torch.int64, device='cuda') for _ in range(nl)]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for k in range(nl):
    lib.mifwt_pyr_profile_buffer(bufs[k].data_ptr())
    ptwt_amd.wavedec2(xs[k % 3], 'db4', level=3)
e1.record()
torch.cuda.synchronize()
lib.mifwt_pyr_profile_buffer(None)
print(f'B={B} dbg={dbg}: {nl} profiled launches, {e0.elapsed_time(e1) / nl * 1e3:.1f} us per call by events')
prev_end = None
for k in range(nl):
    b = bufs[k].view(nwg, 16, 2).cpu()
    used = b[:, 8, 0] > 0
    st, en = b[used, 8, 0].double() / 100.0, b[used, 8, 1].double() / 100.0  # us
    t0 = st.min()
    seg = (b[used, 13, 0] % 134 > 0).long() * 0 + torch.div(b[used, 13, 0] % 134, 33, rounding_mode='floor').clamp(max=3)  # segment of the image from the chunk's first row
    line = f' launch {k}: {int(used.sum())} WGs; starts +0 .. +{st.max() - t0:.1f} us; ends +{en.min() - t0:.1f} .. +{en.max() - t0:.1f} us (median {en.median() - t0:.1f})'
    if prev_end is not None: line += f'; gap after previous launch\'s last end {t0 - prev_end:.1f} us'
    prev_end = en.max()
    print(line)
    if k == nl - 1 and B != 64:
        glo, ghi = b[used, 13, 0], b[used, 13, 1]
        hn = 134
        cls = {}
        for i in range(int(used.sum())):
            lo, hi = int(glo[i]), int(ghi[i])
            units, g = [], lo
            while g < hi:
                r0 = g % hn; n = min(hn - r0, hi - g); units.append(('T' if r0 == 0 else '') + ('B' if r0 + n == hn else '') + str(n)); g += n
            cls.setdefault(' + '.join(units), []).append(float(en[i] - st[i]))
        for key, v in sorted(cls.items(), key=lambda kv: -max(kv[1])):
            print(f'   chunk of units [{key}] (T = top of an image, B = bottom; rows): {len(v)} workgroups, duration mean {sum(v) / len(v):.1f} max {max(v):.1f} us')
    if k == nl - 1:
        for s in range(4):
            m = seg == s
            d = (en - st)[m]
            print(f'   segment {s}: start +{(st[m] - t0).mean():.1f} (max {(st[m] - t0).max():.1f}), duration mean {d.mean():.1f} min {d.min():.1f} max {d.max():.1f}, end mean +{(en[m] - t0).mean():.1f} max +{(en[m] - t0).max():.1f}')
        xcc = b[used, 12, 1] & 15
        dur = (en - st)
        print('   duration by XCC: ' + '  '.join(f'{i}: {dur[xcc == i].mean():.1f}' for i in range(8)))
        print('   duration by image (first 8): ' + '  '.join(f'{dur[torch.arange(len(dur)) // 4 == i].mean():.1f}' for i in range(8)))
        hw = b[used, 12, 0]
        print('   WGs per XCC:', [int((xcc == i).sum()) for i in range(8)], ' distinct (xcc, se, sh, cu):', len(set(zip(xcc.tolist(), ((hw >> 13) & 7).tolist(), ((hw >> 12) & 1).tolist(), ((hw >> 8) & 15).tolist()))))
        roles = {0: 'L1.1', 1: 'L1.0(left edge)', 2: 'L1.3', 3: 'L1.4(right edge, 2 lanes)', 4: 'L1.2', 5: 'L2.0', 6: 'L2.1', 7: 'L2.2', 9: 'L3.0', 10: 'L3.1', 11: 'L3.2'}
        if b[used][:, 12, 0].sum() == 0 and b[used][:, 9, 0].sum() > 0 and b[used][:, 10, 0].sum() == 0:  # the twelve-wave form
            roles = {0: 'L1.0', 1: 'L1.1', 2: 'L1.2', 3: 'L1.3', 4: 'L2.0', 5: 'L2.1', 6: 'L3.0', 7: 'L3.1', 9: 'tail'}
        bb = b[used].double()
        print('   share of a wave\'s cycles spent in barriers: ' + '  '.join(f'{n}: {100 * bb[:, w, 1].sum() / bb[:, w, 0].sum():.0f}%' for w, n in roles.items()))
        sec = bb[:, 14:16, :].reshape(-1, 4)
        l2w = 4 if bb[:, 12, 0].sum() == 0 else 5
        print(f'   first level-2 wave (wave {l2w}): total {bb[:, l2w, 0].mean():.0f} cycles, in barriers {bb[:, l2w, 1].mean():.0f}; window loads {sec[:, 0].mean():.0f}, arithmetic {sec[:, 1].mean():.0f}, ring write + stores {sec[:, 2].mean():.0f}, pad fill {sec[:, 3].mean():.0f}')
        cyc = b[used][:, :5, 0].double()
        print(f'   level-1 wave cycles per WG mean {cyc.mean():.0f} -> {cyc.mean() / (en - st).mean():.0f} cycles per us')
