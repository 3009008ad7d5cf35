"""Does a call loop leave live objects behind (a leak, as opposed to cycles)?  Object counts by type before / after 2000 calls."""
import gc, sys, collections, torch
sys.path.insert(0, '.')
import ptwt_amd
xs = torch.randn(256, 64, 64, device='cuda')
cs = ptwt_amd.wavedec2(xs, 'db2', level=3)
xb = torch.randn(4, 1024, 1024, device='cuda'); cb = ptwt_amd.wavedec2(xb, 'db4', level=3)
for name, fn in {"waverec2 small": lambda: ptwt_amd.waverec2(cs, 'db2'), "wavedec2 small": lambda: ptwt_amd.wavedec2(xs, 'db2', level=3),
                 "waverec2 big": lambda: ptwt_amd.waverec2(cb, 'db4'), "wavedec2 big": lambda: ptwt_amd.wavedec2(xb, 'db4', level=3)}.items():
    for _ in range(50): fn()
    gc.collect()
    before = collections.Counter(type(o).__name__ for o in gc.get_objects())
    c0 = gc.get_count()
    for _ in range(2000): fn()
    c1 = gc.get_count()
    gc.collect()
    after = collections.Counter(type(o).__name__ for o in gc.get_objects())
    diff = {k: after[k] - before[k] for k in after if after[k] - before[k] > 5}
    print(f"{name}: gc counts {c0} -> {c1}; live objects that grew by more than 5 over 2000 calls: {diff}")
