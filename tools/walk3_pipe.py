"""Config 3 (wavedec3 db2 level 3, 8 x 256^3, zero mode): the deep levels of one batch chunk on a side stream beside level 1 of the next
chunk — they are latency-bound launches (8 x 129^3: 44 us, 8 x 66^3: 17 us) that leave most of the chip idle."""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine
xs = [torch.randn(8, 256, 256, 256, device='cuda') for _ in range(3)]
side = torch.cuda.Stream()
W, M = 'db2', 'zero'

def whole(x):
    return ptwt_amd.wavedec3(x, W, mode=M, level=3)

def piped(x, nchunks):
    main = torch.cuda.current_stream()
    keep = []
    for c in x.chunk(nchunks):
        l1 = ptwt_amd.wavedec3(c, W, mode=M, level=1)
        ev = torch.cuda.Event()
        ev.record(main)
        with torch.cuda.stream(side):
            side.wait_event(ev)
            keep.append((l1, ptwt_amd.wavedec3(l1[0], W, mode=M, level=2)))
    main.wait_stream(side)
    return keep

def serial_chunks(x, nchunks):
    return [whole(c) for c in x.chunk(nchunks)]

def t(fn, n=30):
    for i in range(5): fn(xs[i % 3])
    torch.cuda.synchronize()
    r = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n): fn(xs[i % 3])
        e1.record(); torch.cuda.synchronize(); r.append(e0.elapsed_time(e1) / n)
    return sorted(r)[2] * 1e3

print(f"one call, all levels: {t(whole):.1f} us;  level 1 alone: {t(lambda x: ptwt_amd.wavedec3(x, W, mode=M, level=1)):.1f} us", flush=True)
for n in (1, 2, 4, 8):
    print(f"{n} chunks: serial {t(lambda x: serial_chunks(x, n)):.1f} us, deep levels on a side stream {t(lambda x: piped(x, n)):.1f} us", flush=True)
