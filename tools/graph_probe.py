"""Does a whole wavedec2 / waverec2 call capture into a HIP graph (torch.cuda.CUDAGraph) and replay correctly?  Host time per call:
eager enqueue vs graph replay, on a launch-bound shape (small batch) and a multi-launch one (the reference's 2-D speed-test shape)."""
import gc, sys, time, torch
sys.path.insert(0, '.')
import ptwt_amd
for shape, wav, lev, mode in (((16, 64, 64), 'db2', 3, 'reflect'), ((32, 1000, 1000), 'db5', 5, 'periodic'), ((8, 256, 256), 'db4', 4, 'symmetric')):
    x = torch.randn(*shape, device='cuda')
    static_x = x.clone()
    for _ in range(3):
        c = ptwt_amd.wavedec2(static_x, wav, mode=mode, level=lev); y = ptwt_amd.waverec2(c, wav)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            c = ptwt_amd.wavedec2(static_x, wav, mode=mode, level=lev); y = ptwt_amd.waverec2(c, wav)
    torch.cuda.current_stream().wait_stream(s)
    try:
        with torch.cuda.graph(g):
            gc_ = ptwt_amd.wavedec2(static_x, wav, mode=mode, level=lev)
            gy = ptwt_amd.waverec2(gc_, wav)
    except Exception as exc:
        print(shape, 'capture failed:', repr(exc)[:300]); continue
    x2 = torch.randn(*shape, device='cuda')
    static_x.copy_(x2)
    g.replay(); torch.cuda.synchronize()
    ref_c = ptwt_amd.wavedec2(x2, wav, mode=mode, level=lev); ref_y = ptwt_amd.waverec2(ref_c, wav)
    ok = torch.equal(gy, ref_y) and torch.equal(gc_[0], ref_c[0]) and all(torch.equal(a, b) for la, lb in zip(gc_[1:], ref_c[1:]) for a, b in zip(la, lb))
    def timeit(fn, n=300):  # (the cyclic collector off, as bench.py does: a full pass is a 35 ms pause = 100 us per call of a 300-call loop)
        for _ in range(20): fn()
        gc.disable()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n * 1e6
        gc.enable(); return dt
    eager = timeit(lambda: ptwt_amd.waverec2(ptwt_amd.wavedec2(static_x, wav, mode=mode, level=lev), wav))
    graph = timeit(g.replay)
    print(f"{shape} {wav} L{lev} {mode}: graph replay {'bit-identical to eager' if ok else 'DIFFERS'};  wavedec2 + waverec2 per iteration: eager {eager:.1f} us, graph replay {graph:.1f} us", flush=True)
