"""Host-side cost of a transform call WITHOUT a GPU: the launch itself is stubbed out (no kernel runs, outputs stay uninitialised), every
other line of the Python path — checks, folding, plan lookup, allocations, ctypes marshalling, containers — runs as in production.
usage: host_cpu_profile.py [profile]"""
import sys, time, cProfile, pstats
sys.path.insert(0, '.')
import torch, ptwt_amd
from ptwt_amd import _engine
_engine._require_gpu = lambda t: None
_real_empty = torch.empty
_cache = {}
def _fake_empty(*a, **k):  # (allocation cost of big CPU tensors is not what a CUDA caching allocator costs: reuse one tensor per shape)
    key = (a, tuple(sorted((kk, str(v)) for kk, v in k.items())))
    t = _cache.get(key)
    if t is None:
        t = _cache[key] = _real_empty(*a, **k)
    return t
torch.empty = _fake_empty
_engine.HipLevelEngine._run = staticmethod(lambda p, direction, anchor, call, kid=None: None)
CASES = [((4096, 64, 64), 'db2', 3, 'wavedec2'), ((4096, 64, 64), 'db2', 3, 'waverec2'), ((64, 1024, 1024), 'db4', 3, 'wavedec2'), ((64, 1024, 1024), 'db4', 3, 'waverec2'),
         ((32, 1000, 1000), 'db5', 5, 'wavedec2'), ((32, 1000, 1000), 'db5', 5, 'waverec2'), ((8, 64, 64, 64), 'db2', 3, 'wavedec3'), ((32, 100000), 'db5', 10, 'wavedec')]
for shape, wav, lev, fn in CASES:
    x = torch.empty(*shape)
    if 'rec' in fn:
        c = getattr(ptwt_amd, fn.replace('rec', 'dec'))(x, wav, level=lev)
        call = lambda: getattr(ptwt_amd, fn)(c, wav)
    else:
        f = getattr(ptwt_amd, fn)
        call = lambda: f(x, wav, level=lev)
    for _ in range(20): call()
    t0 = time.perf_counter()
    for _ in range(500): call()
    print(f"{fn:9s} {str(shape):20s} {wav} L{lev}: host {1e6 * (time.perf_counter() - t0) / 500:6.1f} us/call")
    if len(sys.argv) > 1 and sys.argv[1] == fn + str(shape[0]):
        pr = cProfile.Profile(); pr.enable()
        for _ in range(2000): call()
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(22)
