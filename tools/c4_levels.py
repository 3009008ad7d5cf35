"""Config 4 (64 x 4096^2 db8 level 4): whole calls both ways and the level-1 / finest-level launches alone, ms."""
import sys, time, torch
sys.path.insert(0, '.')
import ptwt_amd as ptwt
dev = torch.device('cuda:0')
x = [torch.randn(64, 4096, 4096, device=dev) for _ in range(2)]
def timeit(fn, n=10):
    for _ in range(3): fn(0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
c = [ptwt.wavedec2(x[i], 'db8', mode='reflect', level=4) for i in range(2)]
c1 = [ptwt.wavedec2(x[i], 'db8', mode='reflect', level=1) for i in range(2)]
print('wavedec2 L4 %.3f  L1 %.3f   waverec2 L4 %.3f  L1 %.3f' % (
    timeit(lambda i: ptwt.wavedec2(x[i & 1], 'db8', mode='reflect', level=4)), timeit(lambda i: ptwt.wavedec2(x[i & 1], 'db8', mode='reflect', level=1)),
    timeit(lambda i: ptwt.waverec2(c[i & 1], 'db8')), timeit(lambda i: ptwt.waverec2(c1[i & 1], 'db8'))))
