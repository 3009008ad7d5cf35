"""Host side of the small-plane calls (4096 x 64^2 db2 level 3: one launch per call): enqueue time per call, call-loop time, cProfile."""
import cProfile, pstats, sys, time, torch
sys.path.insert(0, '.')
import ptwt_amd
dev = torch.device("cuda:0")
x = torch.randn(4096, 64, 64, device=dev)
c = ptwt_amd.wavedec2(x, 'db2', level=3)
for name, call in (("wavedec2", lambda: ptwt_amd.wavedec2(x, 'db2', level=3)), ("waverec2", lambda: ptwt_amd.waverec2(c, 'db2'))):
    for i in range(20): call()
    torch.cuda.synchronize()
    # enqueue only: a few calls into an empty queue
    enq = []
    for r in range(20):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); call(); enq.append(time.perf_counter() - t0)
    enq.sort()
    t0 = time.perf_counter()
    for i in range(500): call()
    torch.cuda.synchronize()
    loop = (time.perf_counter() - t0) / 500
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record(); call(); e1.record(); torch.cuda.synchronize()
    print(f"{name}: enqueue median {enq[10]*1e6:.1f} us (min {enq[0]*1e6:.1f}), call loop {loop*1e6:.1f} us/call, one call between events {e0.elapsed_time(e1)*1e3:.1f} us")
    pr = cProfile.Profile(); pr.enable()
    for i in range(500): call()
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)
