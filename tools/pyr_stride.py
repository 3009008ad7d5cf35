"""Does the power-of-two image stride (64 x 1024 x 1024 f32: images 4 MiB apart) cost HBM channel conflicts?  The same call on views of
buffers whose images are 1024 + k rows apart."""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
def t(fn, n=50):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    r = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); r.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(r)[3]
for extra_rows, extra_cols in ((0, 0), (1, 0), (3, 0), (8, 0), (17, 0), (0, 16), (0, 32)):
    bufs = [torch.randn(64, 1024 + extra_rows, 1024 + extra_cols, device='cuda') for _ in range(3)]
    xs = [b[:, :1024, :1024] for b in bufs]
    i = [0]
    def f():
        i[0] += 1; return ptwt_amd.wavedec2(xs[i[0] % 3], 'db4', level=3)
    held = [None, None, None]
    def frot():
        i[0] += 1; held[i[0] % 3] = ptwt_amd.wavedec2(xs[i[0] % 3], 'db4', level=3)
    a = t(f); b = t(frot); held[:] = [None] * 3
    print(f'image stride {1024 + extra_rows} rows of {1024 + extra_cols} floats: results dropped {a:.1f} us, rotating outputs {b:.1f} us', flush=True)
    del bufs, xs
    torch.cuda.empty_cache()
