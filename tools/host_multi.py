"""Host side of a multi-launch call (the reference's 2-D speed-test shape: wavedec2 db5 level 5 periodic on 32 x 1000^2 = five per-level
launches; waverec2 = two per-level launches + kernel 22): enqueue time per call and cProfile."""
import cProfile, pstats, sys, time, torch
sys.path.insert(0, '.')
import ptwt_amd
x = torch.randn(32, 1000, 1000, device='cuda')
c = ptwt_amd.wavedec2(x, 'db5', mode='periodic', level=5)
for name, call in (("wavedec2", lambda: ptwt_amd.wavedec2(x, 'db5', mode='periodic', level=5)), ("waverec2", lambda: ptwt_amd.waverec2(c, 'db5'))):
    for i in range(20): call()
    torch.cuda.synchronize()
    enq = []
    for r in range(20):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); call(); enq.append(time.perf_counter() - t0)
    enq.sort()
    t0 = time.perf_counter()
    for i in range(300): call()
    torch.cuda.synchronize()
    loop = (time.perf_counter() - t0) / 300
    print(f"{name}: enqueue median {enq[10]*1e6:.1f} us (min {enq[0]*1e6:.1f}), call loop {loop*1e6:.1f} us/call")
    pr = cProfile.Profile(); pr.enable()
    for i in range(300): call()
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
