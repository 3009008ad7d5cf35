"""The reference's 3-D speed shape (32 x 100^3 db5 periodic, examples/speed_tests/timeitconv_3d.py): whole wavedec3 level 3, each level's
launch alone, and the depth-walk kernel's knobs on level 1 (depth segments, row sub-groups, staged slices)."""
import sys, time, torch
sys.path.insert(0, '.')
import ptwt_amd as ptwt
from ptwt_amd import _engine as E
dev = torch.device('cuda:0')
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
x = torch.randn(B, 100, 100, 100, device=dev)
print('wavedec3 db5 periodic %d x 100^3: level 3 %.1f us' % (B, timeit(lambda: ptwt.wavedec3(x, 'db5', mode='periodic', level=3))))
cur = x
for l in range(3):
    t = timeit(lambda: ptwt.wavedec3(cur, 'db5', mode='periodic', level=1))
    c = ptwt.wavedec3(cur, 'db5', mode='periodic', level=1)
    nb = cur.numel() * 4 + sum(v.numel() for v in c[1].values()) * 4 + c[0].numel() * 4
    print('  level %d: %s -> %s  %.1f us  (%.2f of 8 TB/s; kernel ids %s)' % (l + 1, tuple(cur.shape[1:]), tuple(c[0].shape[1:]), t, nb / t / 8e6, sorted(set(e[0] for e in E.ENGINE.level_events()[-1:])) if hasattr(E.ENGINE, 'level_events') else '?'))
    cur = c[0].contiguous()
f1 = lambda: ptwt.wavedec3(x, 'db5', mode='periodic', level=1)
for seg in (0, 6, 9, 14, 18, 27, 54):
    E.set_option(E.OPT_ROWS_PER_CHUNK, seg)
    row = []
    for nrg in (0, 1, 2):
        E.set_option(E.OPT_PAIR_ROWS, nrg)
        for ahead in (0, 2, 3, 4):
            E.set_option(E.OPT_PREFETCH_PAIRS, ahead)
            row.append('nrg %d ahead %d: %.1f' % (nrg, ahead, timeit(f1, 30)))
    print('  level 1, %2d output slices per segment: ' % seg + '  '.join(row))
