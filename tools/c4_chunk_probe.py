"""Config 4 (64 x 4096^2 db8 level 4): does running the level loop on CHUNKS of the batch (so that the approximations between the
levels are still in the 256 MiB Infinity Cache when the next level reads them) beat one launch per level over the whole batch?
Outputs dropped in both forms (same regime as bench.py's step)."""
import sys, time, torch
sys.path.insert(0, '.')
import ptwt_amd as ptwt

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x = [torch.randn(B, 4096, 4096, device=dev) for _ in range(2)]

def timeit(fn, n=12):
    for _ in range(3): fn(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

def whole(i):
    ptwt.wavedec2(x[i & 1], 'db8', mode='reflect', level=4)

print('whole batch            : %.3f ms' % timeit(whole))
for n in (1, 2, 4, 8, 16, 32):
    def chunked(i, n=n):
        xx = x[i & 1]
        for c in range(0, B, n):
            ptwt.wavedec2(xx[c:c + n], 'db8', mode='reflect', level=4)
    print('chunks of %2d images    : %.3f ms' % (n, timeit(chunked)))
# levels 1-2 chunked, the rest on the whole batch is not expressible through the public call: see tools/c4_chunk_probe2.py
