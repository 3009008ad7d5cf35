"""Config 3 (wavedec3 db2 level 3 on 8 x 256^3, zero mode) and the reference's 3-D shape (32 x 100^3 db5 periodic): the depth-walking
analysis kernel (tile mode 4) against the default routes; per-level times, depth-segment and staging-depth sweeps."""
import sys, torch
sys.path.insert(0, '.')
import ptwt_amd
from ptwt_amd import _engine

def t(fn, args, n=30):
    for i in range(5): fn(args[i % len(args)])
    torch.cuda.synchronize()
    r = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n): fn(args[i % len(args)])
        e1.record(); torch.cuda.synchronize(); r.append(e0.elapsed_time(e1) / n)
    return sorted(r)[2] * 1e3

def run(label, shape, wavelet, mode, levels):
    xs = [torch.randn(*shape, device='cuda') for _ in range(3)]
    for tm, name in ((1, "bricks / composed"), (4, "walk")):
        _engine.set_option(_engine.OPT_TILE_MODE, tm)
        try:
            row = [f"{t(lambda x: ptwt_amd.wavedec3(x, wavelet, mode=mode, level=l), xs):.1f}" for l in levels]
            print(f"{label} {name}: levels {levels} -> {row} us", flush=True)
        finally:
            _engine.set_option(_engine.OPT_TILE_MODE, 0)
    return xs

if __name__ == '__main__':
    xs = run("config 3", (8, 256, 256, 256), "db2", "zero", (1, 3))
    _engine.set_option(_engine.OPT_TILE_MODE, 4)
    for seg in (0, 8, 11, 13, 17, 22, 33, 65, 129):
        _engine.set_option(_engine.OPT_ROWS_PER_CHUNK, seg)
        print(f"  walk level 1, {seg or 'auto'} output slices per segment: {t(lambda x: ptwt_amd.wavedec3(x, 'db2', mode='zero', level=1), xs):.1f} us", flush=True)
    _engine.set_option(_engine.OPT_ROWS_PER_CHUNK, 0)
    for pf in (2, 3, 4, 5, 6):
        _engine.set_option(_engine.OPT_PREFETCH_PAIRS, pf)
        print(f"  walk level 1, {pf} slices ahead: {t(lambda x: ptwt_amd.wavedec3(x, 'db2', mode='zero', level=1), xs):.1f} us", flush=True)
    _engine.set_option(_engine.OPT_PREFETCH_PAIRS, 0)
    for dbg, name in ((1, "no stores"), (2, "no loads"), (3, "neither"), (4, "no W / H pass"), (7, "barriers + D pass only"), (8, "segments outermost")):
        _engine.set_option(_engine.OPT_DEBUG, dbg)
        print(f"  walk level 1, {name}: {t(lambda x: ptwt_amd.wavedec3(x, 'db2', mode='zero', level=1), xs):.1f} us", flush=True)
    _engine.set_option(_engine.OPT_DEBUG, 0)
    for seg, pf in ((129, 4), (129, 6), (65, 6), (43, 6), (0, 6)):
        _engine.set_option(_engine.OPT_ROWS_PER_CHUNK, seg); _engine.set_option(_engine.OPT_PREFETCH_PAIRS, pf)
        print(f"  walk level 1, {seg or 'auto'} slices per segment, {pf} ahead: {t(lambda x: ptwt_amd.wavedec3(x, 'db2', mode='zero', level=1), xs):.1f} us", flush=True)
    _engine.set_option(_engine.OPT_ROWS_PER_CHUNK, 0); _engine.set_option(_engine.OPT_PREFETCH_PAIRS, 0)
    _engine.set_option(_engine.OPT_NT_STORE, 1)
    print(f"  walk level 1, nt stores: {t(lambda x: ptwt_amd.wavedec3(x, 'db2', mode='zero', level=1), xs):.1f} us", flush=True)
    _engine.set_option(_engine.OPT_NT_STORE, 0)
    _engine.set_option(_engine.OPT_TILE_MODE, 0)
    del xs
    run("reference 3-D shape", (32, 100, 100, 100), "db5", "periodic", (1, 3))
    run("db4 128^3 x 16", (16, 128, 128, 128), "db4", "reflect", (1, 3))

