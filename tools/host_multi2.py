import sys, time, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
x = torch.randn(32, 1000, 1000, device='cuda')
cs = [ptwt_amd.wavedec2(torch.randn(32, 1000, 1000, device='cuda'), 'db5', mode='periodic', level=5) for _ in range(3)]
def loop(n, fn):
    for i in range(30): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for i in range(n): fn(i)
    t_enq = time.perf_counter() - t0
    e1.record(); torch.cuda.synchronize()
    return t_enq / n * 1e6, (time.perf_counter() - t0) / n * 1e6, e0.elapsed_time(e1) / n * 1e3
print("waverec2 same set     : enqueue %.1f wall %.1f gpu-span %.1f us" % loop(300, lambda i: ptwt_amd.waverec2(cs[0], 'db5')))
print("waverec2 rotating sets: enqueue %.1f wall %.1f gpu-span %.1f us" % loop(300, lambda i: ptwt_amd.waverec2(cs[i % 3], 'db5')))
print("wavedec2              : enqueue %.1f wall %.1f gpu-span %.1f us" % loop(300, lambda i: ptwt_amd.wavedec2(x, 'db5', mode='periodic', level=5)))
_engine.level_events = []
for i in range(5): ptwt_amd.waverec2(cs[0], 'db5')
torch.cuda.synchronize()
ev, _engine.level_events = _engine.level_events, None
print([(k, tuple(ext), round(s.elapsed_time(e) * 1e3, 1)) for (tag, k, ext, s, e) in ev[-3:]])
print([tuple(t.stride()) for t in cs[0][1]], [tuple(t.shape) for t in cs[0][1]])
for k in range(3):
    _engine.level_events = []
    for i in range(6): ptwt_amd.waverec2(cs[k], 'db5')
    torch.cuda.synchronize()
    ev, _engine.level_events = _engine.level_events, None
    print("set", k, [(kk, tuple(ext), round(s.elapsed_time(e) * 1e3, 1)) for (tag, kk, ext, s, e) in ev[-3:]], [hex(t.data_ptr() % 4096) for t in cs[k][-1]], [t.data_ptr() // (1 << 20) for t in (cs[k][0], cs[k][-1][0])])
def loop2(n, fn):
    for i in range(30): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for k in range(3):
    print("only set", k, "%.1f us" % loop2(200, lambda i: ptwt_amd.waverec2(cs[k], 'db5')))
print("sets 0,1 alternating %.1f us" % loop2(200, lambda i: ptwt_amd.waverec2(cs[i % 2], 'db5')))
print("sets 0,1,2 rotating %.1f us" % loop2(300, lambda i: ptwt_amd.waverec2(cs[i % 3], 'db5')))
print("sets 0,2 alternating %.1f us" % loop2(300, lambda i: ptwt_amd.waverec2(cs[2 * (i % 2)], 'db5')))
_engine.level_events = []
for i in range(12): ptwt_amd.waverec2(cs[i % 3], 'db5')
torch.cuda.synchronize()
ev, _engine.level_events = _engine.level_events, None
print("rotating, per launch:", [(kk, tuple(ext), round(s.elapsed_time(e) * 1e3, 1)) for (tag, kk, ext, s, e) in ev[-9:]])
xs = [torch.randn(32, 1000, 1000, device='cuda') for _ in range(3)]
print("wavedec2 rotating inputs %.1f us" % loop2(300, lambda i: ptwt_amd.wavedec2(xs[i % 3], 'db5', mode='periodic', level=5)))
