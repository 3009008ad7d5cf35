import os, sys, time
sys.path.insert(0, '.')
import torch, ptwt_amd
dev = torch.device("cuda:0")
import cProfile, pstats
for shape, wav, lev, fn in [((256, 256, 256), 'db4', 3, 'wavedec2'), ((256, 256, 256), 'db4', 3, 'waverec2'), ((8, 64, 64, 64), 'db2', 2, 'wavedec3')]:
    x = torch.randn(*shape, device=dev)
    if fn == 'waverec2':
        c = ptwt_amd.wavedec2(x, wav, level=lev)
        call = lambda: ptwt_amd.waverec2(c, wav)
    else:
        f = getattr(ptwt_amd, fn)
        call = lambda: f(x, wav, level=lev)
    for i in range(5): call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(200): call()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"{fn} {shape}: enqueue {1e6*(t1-t0)/200:.1f} us/call")
    pr = cProfile.Profile(); pr.enable()
    for i in range(300): call()
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
