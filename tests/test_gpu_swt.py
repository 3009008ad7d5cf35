"""GPU tests (``-m gpu``) of the stationary transform ``swt`` / ``iswt`` against golden vectors of the reference's
own functions (tests/golden/ptwt_ref_swt.npz: coefficients, reconstruction and autograd gradients; fp64, 1e-12 /
1e-11 norm-wise) plus fp32 / fp16 round trips at sizes with many segments per row."""
import numpy as np
import pytest
import torch

import ptwt_amd
from tests import _golden as G

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def weight(t, i):
    return torch.cos(0.37 * torch.arange(t.numel(), dtype=torch.float64, device=t.device) + i).reshape(t.shape).to(t.dtype)


def test_swt_vs_reference_goldens():
    z, idx = G.load("ptwt_ref_swt.npz")
    for case in idx:
        k = case["key"]
        x = torch.from_numpy(z[k + "_x"]).to(dev()).requires_grad_(True)
        c = ptwt_amd.swt(x, case["wavelet"], case["level"], **case["kw"])
        assert len(c) == case["ncoef"]
        for i, t in enumerate(c):
            want = z["%s_c%d" % (k, i)]
            assert tuple(t.shape) == want.shape
            assert G.relerr(t.detach().cpu().numpy(), want) < 1e-12, (case, i)
        (gx,) = torch.autograd.grad(sum((weight(t, i) * t).sum() for i, t in enumerate(c)), x)
        assert G.relerr(gx.cpu().numpy(), z[k + "_gx"]) < 1e-11, (case, "swt backward")
        leaves = [t.detach().clone().requires_grad_(True) for t in c]
        y = ptwt_amd.iswt(leaves, case["wavelet"], **case["kw"])
        assert G.relerr(y.detach().cpu().numpy(), z[k + "_rec"]) < 1e-12, (case, "iswt")
        gl = torch.autograd.grad((weight(y, 7) * y).sum(), leaves)
        for i, g in enumerate(gl):
            assert G.relerr(g.cpu().numpy(), z["%s_gc%d" % (k, i)]) < 1e-11, (case, "iswt backward", i)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-6), (torch.float64, 1e-12)])
def test_swt_roundtrip_large(dtype, tol):
    x = torch.randn(7, 40960, device=dev(), dtype=dtype)
    for wavelet in ("haar", "db4", "db10", "db12", "db20"):  # the last two: run-time tap loop (> 20 taps)
        c = ptwt_amd.swt(x, wavelet, 5)
        assert all(t.shape == x.shape for t in c) and len(c) == 6
        y = ptwt_amd.iswt(c, wavelet)
        assert G.relerr(y.cpu().numpy(), x.cpu().numpy()) < tol, wavelet


def test_swt_half_storage_and_errors():
    x = torch.randn(3, 4096, device=dev())
    with pytest.raises(ValueError):
        ptwt_amd.swt(x.half(), "db2", 2)
    ptwt_amd.set_half_storage(True)
    try:
        c16 = ptwt_amd.swt(x.half(), "db2", 3)
        c32 = ptwt_amd.swt(x.half().float(), "db2", 3)
        for a, b in zip(c16, c32):
            assert a.dtype == torch.float16
            assert G.relerr(a.float().cpu().numpy(), b.cpu().numpy()) < 1e-3
    finally:
        ptwt_amd.set_half_storage(False)
    with pytest.raises(RuntimeError):
        ptwt_amd.swt(torch.randn(2, 64), "db2", 2)  # CPU tensor: no fallback
    assert ptwt_amd.stationary_transform.swt_max_level(96) == 5
