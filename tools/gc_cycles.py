"""Which reference cycles does a call leave behind?  (They cost nothing until the collector's full pass: ~35 ms with torch loaded.)"""
import gc, sys, collections, torch
sys.path.insert(0, '.')
import ptwt_amd
x = torch.randn(64, 64, 64, device='cuda')
cs = ptwt_amd.wavedec2(x, 'db2', level=3)
x1 = torch.randn(8, 100000, device='cuda'); c1 = ptwt_amd.wavedec(x1, 'db5', level=6, mode='periodic')
xb = torch.randn(4, 1024, 1024, device='cuda'); cb = ptwt_amd.wavedec2(xb, 'db4', level=3)
x3 = torch.randn(2, 64, 64, 64, device='cuda'); c3 = ptwt_amd.wavedec3(x3, 'db2', level=2)
calls = {"wavedec2 small": lambda: ptwt_amd.wavedec2(x, 'db2', level=3), "waverec2 small": lambda: ptwt_amd.waverec2(cs, 'db2'),
         "wavedec": lambda: ptwt_amd.wavedec(x1, 'db5', level=6, mode='periodic'), "waverec": lambda: ptwt_amd.waverec(c1, 'db5'),
         "wavedec2 big": lambda: ptwt_amd.wavedec2(xb, 'db4', level=3), "waverec2 big": lambda: ptwt_amd.waverec2(cb, 'db4'),
         "wavedec3": lambda: ptwt_amd.wavedec3(x3, 'db2', level=2), "waverec3": lambda: ptwt_amd.waverec3(c3, 'db2'),
         "fswavedec2": lambda: ptwt_amd.fswavedec2(xb, 'db4', level=3), "fswaverec2": lambda: ptwt_amd.fswaverec2(ptwt_amd.fswavedec2(xb, 'db4', level=3), 'db4')}
for name, fn in calls.items():
    for _ in range(3): fn()
    gc.collect()
    gc.set_debug(gc.DEBUG_SAVEALL)
    for _ in range(5): fn()
    n = gc.collect()
    gc.set_debug(0)
    kinds = collections.Counter(type(o).__name__ for o in gc.garbage)
    detail = ''
    for o in gc.garbage:
        if type(o).__name__ == 'function':
            detail += f" fn:{o.__qualname__}"
        if type(o).__name__ == 'cell':
            try: detail += f" cell->{type(o.cell_contents).__name__}"
            except ValueError: pass
    print(f"{name}: {n} unreachable after 5 calls: {dict(kinds)}{detail[:300]}")
    gc.garbage.clear()
