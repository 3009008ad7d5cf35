"""Out-of-bounds canaries on the GPU ("canaries around outputs").  Every kernel of libmifwt.so writes through raw
pointers with hand-computed offsets into buffers the ctypes binding (`_engine.py`) allocates with ``torch.empty``.  Here every such
allocation — level buffers, reconstruction outputs, gradient buffers, scratch workspaces — is carved out of a larger block filled with a
byte pattern, at a shifted start, and after each call (a) every guard byte around the carved region must still hold the pattern,
(b) the inputs must be unmodified, (c) the results must equal those of the same call on ordinary allocations bit for bit.  The
scenarios walk every kernel id the dispatcher knows (`_engine.level_events` pins which ran); odd extents, last images of a batch,
strided inputs, offsets beyond 2^31 bytes."""
import numpy as np
import pytest
import torch

import ptwt_amd
from ptwt_amd import _engine
from tests import _golden as G

pytestmark = pytest.mark.gpu

GUARD = 4096  # bytes on either side
PATTERN = 0xA5


class _GuardedTorch:
    """Stands in for the ``torch`` module inside ``_engine``: ``empty`` on a real device carves the tensor out of a guarded block."""

    def __init__(self):
        self.blocks = []
        self.shift = 0

    def __getattr__(self, name):
        return getattr(torch, name)

    def empty(self, *shape, dtype=None, device=None, **kw):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
            shape = tuple(shape[0])
        dev = torch.device(device) if device is not None else torch.device("cpu")
        if dev.type != "cuda":
            return torch.empty(shape, dtype=dtype, device=device, **kw)
        dtype = dtype or torch.float32
        esize = torch.empty(0, dtype=dtype).element_size()
        n = int(np.prod(shape)) if len(shape) else 1
        self.shift = (self.shift + 1) % 4
        lead = GUARD + 256 * self.shift  # (starts stay 256-byte aligned, as the caching allocator's blocks are)
        raw = torch.full((lead + n * esize + GUARD,), PATTERN, dtype=torch.uint8, device=dev)
        view = raw[lead : lead + n * esize].view(dtype).view(shape)
        self.blocks.append((raw, lead, n * esize))
        return view

    def zeros(self, *shape, dtype=None, device=None, **kw):
        dev = torch.device(device) if device is not None else torch.device("cpu")
        if dev.type != "cuda":
            return torch.zeros(*shape, dtype=dtype, device=device, **kw)
        t = self.empty(*shape, dtype=dtype, device=device)
        t.zero_()
        return t

    def check(self, what):
        torch.cuda.synchronize()
        for raw, lead, nbytes in self.blocks:
            head, tail = raw[:lead], raw[lead + nbytes :]
            assert bool((head == PATTERN).all()), f"{what}: bytes BEFORE a {nbytes}-byte allocation were written"
            assert bool((tail == PATTERN).all()), f"{what}: bytes AFTER a {nbytes}-byte allocation were written"
        n = len(self.blocks)
        self.blocks.clear()
        return n


@pytest.fixture()
def guarded(monkeypatch):
    g = _GuardedTorch()
    from ptwt_amd import _fwt, stationary_transform

    for mod in (_engine, _fwt, stationary_transform):  # (every module of the package that allocates what a kernel writes)
        monkeypatch.setattr(mod, "torch", g)
    yield g


def dev():
    return torch.device("cuda:0")


def _flat(c):
    return [t for _, t in G.flatten_coeffs(c)] if not isinstance(c, torch.Tensor) else [c]


SEEN = set()


def _run(guarded, what, fn, inputs):
    """fn(*inputs) on guarded allocations: guards, inputs, and equality with the unguarded call."""
    keep = [t.clone() for t in inputs]
    _engine.level_events = []
    try:
        got = fn(*inputs)
        n = guarded.check(what)
        SEEN.update(e[1] for e in _engine.level_events)
    finally:
        _engine.level_events = None
    assert n > 0, what
    for a, b in zip(inputs, keep):
        assert torch.equal(a, b), f"{what}: an input was modified"
    from ptwt_amd import _fwt, stationary_transform

    mods = (_engine, _fwt, stationary_transform)
    for mod in mods:
        mod.torch = torch
    try:
        ref = fn(*inputs)
    finally:
        for mod in mods:
            mod.torch = guarded
    for a, b in zip(_flat(got), _flat(ref)):
        assert torch.equal(a, b) or (torch.isnan(a) == torch.isnan(b)).all(), what
    return got


def _x(*shape, dtype=torch.float32, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g).to(dtype).to(dev())


CASES_2D = [
    # (shape, wavelet, level, mode, dtype, options)
    ((3, 269, 1031), "db4", 3, "reflect", torch.float32, {}),             # 16: streaming analysis, odd extents; 22 back
    ((2, 131, 523), "db2", 2, "zero", torch.float32, {}),                 # 16 / 22 with two levels
    ((5, 200, 300), "db3", 3, "symmetric", torch.float32, {}),            # 12 / 13: two levels per launch + a per-level one
    ((7, 61, 67), "db2", 3, "periodic", torch.float32, {12: 3}),          # 20 / 21: whole pyramid of a small plane (forced on a small batch)
    ((2, 97, 131), "sym5", 2, "constant", torch.float32, {}),             # 7 / 8: tile kernels (10 taps)
    ((2, 90, 150), "db4", 2, "reflect", torch.float64, {}),               # 7 / 8 in double
    ((1, 1600, 1536), "db8", 2, "reflect", torch.float32, {}),            # 1 / 2: streaming wave strips (16 taps, big plane)
    ((1, 1100, 1100), "db3", 1, "symmetric", torch.float32, {}),          # 2: synthesis stream L > 4 on >= 1024^2 / single-level 16, 22
    ((2, 70, 90), "coif17", 1, "zero", torch.float32, {}),                # 0: generic (102 taps)
    ((2, 160, 420), "db10", 2, "reflect", torch.float32, {}),             # tile kernels, 20 taps
]


@pytest.mark.parametrize("case", CASES_2D, ids=lambda c: f"{c[1]}-{'x'.join(map(str, c[0]))}-{c[3]}-{str(c[4]).split('.')[-1]}")
def test_canaries_2d(guarded, case):
    shape, wavelet, level, mode, dtype, opts = case
    for k, v in opts.items():
        _engine.set_option(k, v)
    try:
        x = _x(*shape, dtype=dtype)
        c = _run(guarded, f"wavedec2 {case}", lambda t: ptwt_amd.wavedec2(t, wavelet, mode=mode, level=level), [x])
        flat = _flat(c)
        _run(guarded, f"waverec2 {case}", lambda *ts: ptwt_amd.waverec2((ts[0], *[tuple(ts[1 + 3 * k : 4 + 3 * k]) for k in range(level)]), wavelet), flat)
        cs = _run(guarded, f"fswavedec2 {case}", lambda t: ptwt_amd.fswavedec2(t, wavelet, mode=mode, level=level), [x])
        _run(guarded, f"fswaverec2 {case}", lambda *ts: ptwt_amd.fswaverec2((ts[0], *[dict(zip(("ad", "da", "dd"), ts[1 + 3 * k : 4 + 3 * k])) for k in range(level)]), wavelet),
             [cs[0]] + [d[k] for d in cs[1:] for k in ("ad", "da", "dd")])
    finally:
        for k in opts:
            _engine.set_option(k, 0)


def test_canaries_strided_inputs_and_last_image(guarded):
    """Non-default axes (permuted views), a batch slice that ends at the last image, a column-strided view: nothing is copied in front of
    the kernels for these (descriptor strides), so the guards see the kernels' own addressing."""
    big = _x(4, 300, 3, 310)
    x = big.permute(0, 2, 1, 3)  # axes (-2, -1) strided by 3 * 310 / 1
    _run(guarded, "wavedec2 permuted", lambda t: ptwt_amd.wavedec2(t, "db4", mode="reflect", level=2), [x])
    _run(guarded, "wavedec2 axes=(1, 3)", lambda t: ptwt_amd.wavedec2(t, "db2", mode="zero", level=2, axes=(1, 3)), [big])
    wide = _x(3, 200, 1200)
    _run(guarded, "wavedec2 column view", lambda t: ptwt_amd.wavedec2(t[:, 3:197, 5:1100], "db4", mode="symmetric", level=3), [wide])
    _run(guarded, "wavedec2 last image", lambda t: ptwt_amd.wavedec2(t[2:], "db4", mode="reflect", level=3), [wide])


CASES_1D = [
    ((3, 40001), "db5", 6, "periodic", torch.float32),   # 17 / 18 long rows + 14 / 15 tails
    ((5, 4097), "db4", 4, "reflect", torch.float32),     # 14 / 15
    ((2, 3001), "sym8", 3, "symmetric", torch.float64),  # f64 tails / streaming passes
    ((7, 999), "haar", 1, "zero", torch.float32),        # 3 / 4: one level
    ((600, 700), "db2", 3, "reflect", torch.float32),    # 15: many short rows, the coarse synthesis levels in one launch
    ((2, 513), "coif17", 1, "constant", torch.float32),  # 0
]


@pytest.mark.parametrize("case", CASES_1D, ids=lambda c: f"{c[1]}-{'x'.join(map(str, c[0]))}-{c[3]}")
def test_canaries_1d(guarded, case):
    shape, wavelet, level, mode, dtype = case
    x = _x(*shape, dtype=dtype)
    c = _run(guarded, f"wavedec {case}", lambda t: ptwt_amd.wavedec(t, wavelet, mode=mode, level=level), [x])
    _run(guarded, f"waverec {case}", lambda *ts: ptwt_amd.waverec(list(ts), wavelet), list(c))


CASES_3D = [
    ((2, 161, 162, 163), "db2", 2, "zero", torch.float32, {}),       # 24 / 25 walk on the big level, bricks (9 / 10) below
    ((2, 45, 47, 49), "db3", 2, "reflect", torch.float32, {}),       # 9 / 10 bricks
    ((3, 52, 60, 70), "db5", 2, "periodic", torch.float32, {}),      # 5 / 6 composed
    ((2, 130, 131, 132), "db4", 1, "symmetric", torch.float32, {}),  # 24 / 25 with eight taps
    ((2, 40, 41, 42), "db2", 2, "zero", torch.float64, {}),          # f64: 24 / 25 (two rows per workgroup), then the composed route on 21 x 22 x 22
    ((1, 70, 67, 131), "db3", 2, "symmetric", torch.float64, {}),    # f64: 24 / 25, odd extents, two 1-KiB requests per staged row
    ((1, 128, 66, 250), "db2", 1, "periodic", torch.float64, {}),    # f64: 24 / 25, four rows per workgroup (>= 2^20 samples)
    ((24, 40, 50, 60), "db5", 1, "periodic", torch.float32, {}),     # 24 in its SLAB form by the default route (ten taps, 24 volumes of 1.2e5 samples)
    ((2, 30, 101, 99), "db5", 1, "symmetric", torch.float32, {_engine.OPT_TILE_MODE: 4}),  # slab form: two slabs a volume, the second ragged, odd extents
    ((3, 17, 23, 128), "db4", 1, "reflect", torch.float32, {_engine.OPT_TILE_MODE: 4}),    # slab form: eight taps, one row per request
]


@pytest.mark.parametrize("case", CASES_3D, ids=lambda c: f"{c[1]}-{'x'.join(map(str, c[0]))}-{c[3]}-{str(c[4]).split('.')[-1]}")
def test_canaries_3d(guarded, case):
    shape, wavelet, level, mode, dtype, opts = case
    x = _x(*shape, dtype=dtype)
    for key, value in opts.items():
        _engine.set_option(key, value)
    try:
        c = _run(guarded, f"wavedec3 {case}", lambda t: ptwt_amd.wavedec3(t, wavelet, mode=mode, level=level), [x])
    finally:
        for key in opts:
            _engine.set_option(key, 0)
    keys = list(c[1].keys())
    flat = [c[0]] + [d[k] for d in c[1:] for k in keys]
    _run(guarded, f"waverec3 {case}", lambda *ts: ptwt_amd.waverec3((ts[0], *[dict(zip(keys, ts[1 + 7 * k : 8 + 7 * k])) for k in range(level)]), wavelet), flat)


def test_canaries_half_storage_matrix_core_kernels(guarded):
    """ids 11 / 23 (f16 storage, 18-32 taps) and the vector f16 tile kernels."""
    with ptwt_amd.half_storage():
        for shape, wavelet, level in (((2, 300, 420), "sym16", 2), ((2, 200, 260), "db12", 1), ((3, 130, 170), "db4", 2)):
            x = _x(*shape, dtype=torch.float16)
            c = _run(guarded, f"fswavedec2 f16 {shape} {wavelet}", lambda t: ptwt_amd.fswavedec2(t, wavelet, level=level), [x])
            _run(guarded, f"fswaverec2 f16 {shape} {wavelet}", lambda *ts: ptwt_amd.fswaverec2((ts[0], *[dict(zip(("ad", "da", "dd"), ts[1 + 3 * k : 4 + 3 * k])) for k in range(level)]), wavelet),
                 [c[0]] + [d[k] for d in c[1:] for k in ("ad", "da", "dd")])


def test_canaries_gradients_packets_swt(guarded):
    """Adjoint kernels (incl. the border kernel), tap correlations, packet levels, stationary levels."""
    x = _x(3, 150, 1040).requires_grad_(True)
    for mode in ("reflect", "zero"):
        def fwd_bwd(t):
            outs = _flat(ptwt_amd.wavedec2(t, "db4", mode=mode, level=3))
            return torch.autograd.grad(outs, t, [torch.ones_like(o) for o in outs])[0]
        _run(guarded, f"wavedec2 backward {mode}", fwd_bwd, [x])
    taps = ptwt_amd._wavelets.as_wavelet("db3").filter_bank
    tb = tuple(torch.tensor(t, dtype=torch.float32, device=dev(), requires_grad=True) for t in taps)
    def learn(t):
        outs = _flat(ptwt_amd.wavedec2(t, tb, mode="symmetric", level=2))
        return torch.stack(torch.autograd.grad(outs, list(tb), [torch.ones_like(o) for o in outs], allow_unused=True)[:2])
    _run(guarded, "tap gradients", learn, [x.detach()[:, :90, :200].contiguous()])
    p = _x(4, 64, 96)
    _run(guarded, "WaveletPacket2D", lambda t: torch.stack([ptwt_amd.WaveletPacket2D(t, "db2", mode="reflect", maxlevel=2)[k] for k in ("aa", "dd")]), [p])
    s = _x(3, 256)
    _run(guarded, "swt / iswt", lambda t: ptwt_amd.iswt(ptwt_amd.swt(t, "db3", level=3), "db3"), [s])


def test_canaries_offsets_beyond_two_gib(guarded):
    """Level buffers and inputs larger than 2^31 bytes (kernel 16 + 22, the last image's offsets do not fit 32 bits)."""
    x = torch.randn(560, 1024, 1024, device=dev())
    c = _run(guarded, "wavedec2 560 x 1024^2", lambda t: ptwt_amd.wavedec2(t, "db4", level=3), [x])
    del c
    torch.cuda.empty_cache()


def test_canaries_walked_every_kernel_id():
    """(runs last in this module) the scenarios above reached every kernel family of the dispatcher."""
    want = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 20, 21, 22, 23, 24, 25}
    missing = want - SEEN
    assert not missing, f"kernel ids not reached by the canary scenarios: {sorted(missing)} (reached {sorted(SEEN)})"
