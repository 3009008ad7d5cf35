"""Whole calls on a set of shapes — one library per process (MIFWT_LIB; MIFWT_ALLOW_ABI_MISMATCH=1 for the round-5 build): us per call,
results dropped.  Run once per library and compare (tools/: A/B of rounds)."""
import os, sys, time, torch, warnings
warnings.simplefilter('ignore')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptwt_amd
dev = torch.device('cuda:0')
def t(fn, n=60):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
cases = [("wavedec2", (64, 1024, 1024), w, 3, m) for w, m in (("haar", "reflect"), ("db2", "reflect"), ("db3", "symmetric"), ("db4", "zero"), ("db4", "constant"))]
cases += [("wavedec2", (64, 1024, 1024), "db4", 2, "reflect"), ("wavedec2", (64, 1024, 1024), "db4", 1, "reflect"), ("wavedec2", (16, 2048, 2048), "db4", 3, "reflect"),
          ("wavedec2", (65, 1024, 1024), "db4", 3, "reflect"), ("wavedec2", (100, 1000, 1000), "db3", 3, "reflect"), ("wavedec2", (32, 1000, 1000), "db5", 5, "periodic"),
          ("wavedec2", (256, 512, 512), "db2", 3, "reflect"), ("waverec2", (64, 1024, 1024), "db4", 3, "reflect"), ("wavedec3", (8, 256, 256, 256), "db2", 3, "zero"),
          ("wavedec", (32, 1000000), "db5", 10, "periodic")]
out = []
for fn, shape, w, lev, mode in cases:
    x = torch.randn(*shape, device=dev)
    if fn == "waverec2":
        c = ptwt_amd.wavedec2(x, w, mode=mode, level=lev); f = lambda: ptwt_amd.waverec2(c, w)
    else:
        g = getattr(ptwt_amd, fn); f = lambda: g(x, w, mode=mode, level=lev)
    out.append("%-9s %-22s %-5s L%d %-9s %8.1f" % (fn, "x".join(map(str, shape)), w, lev, mode, t(f)))
    del x
print("\n".join(out))
