"""GPU tests (``-m gpu``) of HIP-graph capture of whole calls (`ptwt_amd.capture`): replays are bit-identical to eager calls on fresh
data for every function family and route (multi-level launches, per-level kernels, 1-D / 3-D, f16 storage), containers keep their types,
geometry mismatches and host-synchronising calls are refused."""
import numpy as np
import pytest
import torch

import ptwt_amd
from tests import _golden as G

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def flat(coeffs):
    return [t for _, t in G.flatten_coeffs(coeffs)]


@pytest.mark.parametrize("fn,rec,shape,wavelet,level,mode", [
    ("wavedec2", "waverec2", (16, 64, 64), "db2", 3, "reflect"),          # small-plane kernels 20 / 21
    ("wavedec2", "waverec2", (4, 600, 520), "db4", 3, "symmetric"),       # streaming multi-level kernels 16 / 22
    ("wavedec2", "waverec2", (3, 300, 260), "db5", 4, "periodic"),        # per-level tile kernels
    ("fswavedec2", "fswaverec2", (2, 129, 200), "db3", 2, "constant"),
    ("wavedec", "waverec", (5, 30000), "db5", 8, "periodic"),             # 1-D multi-level kernels
    ("wavedec3", "waverec3", (2, 40, 36, 44), "db2", 2, "zero"),
])
def test_captured_calls_replay_bit_identically(fn, rec, shape, wavelet, level, mode):
    torch.manual_seed(3)
    f, r = getattr(ptwt_amd, fn), getattr(ptwt_amd, rec)
    x = torch.randn(*shape, device=dev())
    fwd = ptwt_amd.capture(lambda t: f(t, wavelet, mode=mode, level=level), x)
    both = ptwt_amd.capture(lambda t: r(f(t, wavelet, mode=mode, level=level), wavelet), x)
    for seed in (4, 5):
        torch.manual_seed(seed)
        x2 = torch.randn(*shape, device=dev())
        want = f(x2, wavelet, mode=mode, level=level)
        got = fwd(x2)
        assert type(got) is type(want) and len(got) == len(want)
        for a, b in zip(flat(got), flat(want)):
            assert torch.equal(a, b)
        for ga, wa in zip(got[1:], want[1:]):
            assert type(ga) is type(wa)
        assert torch.equal(both(x2), r(want, wavelet))
    kept = fwd.cloned(x)
    fwd(x2)
    for a, b in zip(flat(kept), flat(f(x, wavelet, mode=mode, level=level))):
        assert torch.equal(a, b)  # cloned outputs survive the next replay


def test_capture_refuses_other_geometries_and_half_storage_works():
    x = torch.randn(4, 128, 128, device=dev())
    fwd = ptwt_amd.capture(lambda t: ptwt_amd.wavedec2(t, "db2", level=2), x)
    with pytest.raises(ValueError, match="captured for"):
        fwd(torch.randn(4, 128, 130, device=dev()))
    with pytest.raises(ValueError, match="captured for"):
        fwd(x.double())
    with pytest.raises(RuntimeError):
        ptwt_amd.capture(lambda t: t, torch.randn(4, 4))
    ptwt_amd.set_half_storage(True)
    try:
        xh = torch.randn(2, 300, 402, device=dev()).half()
        g = ptwt_amd.capture(lambda t: ptwt_amd.fswavedec2(t, "sym16", level=2), xh)  # matrix-core kernels
        x2 = torch.randn(2, 300, 402, device=dev()).half()
        for a, b in zip(flat(g(x2)), flat(ptwt_amd.fswavedec2(x2, "sym16", level=2))):
            assert torch.equal(a, b)
    finally:
        ptwt_amd.set_half_storage(False)
