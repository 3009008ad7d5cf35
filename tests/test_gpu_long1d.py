"""GPU parity tests (``-m gpu``) of ``mifwt_dwt1_fwd_long`` (kernel id 17: several 1-D analysis levels of long rows per launch,
a chunk per workgroup, mifwt_dwt1_long.hip) against the fp64 numpy oracle and the per-level kernels.

Tolerance: fp32 <= 1e-6 norm-wise per sub-band vs the fp64 oracle.  Every case asserts that the chunked kernel
ran (``_engine.level_events``), so a silent per-level fallback cannot pass.
"""
import numpy as np
import pytest
import torch

import ptwt_amd
from oracle import fwt_oracle as O
from ptwt_amd import _engine
from tests import _golden as G

pytestmark = pytest.mark.gpu

TOL32 = 1e-6
MODES = ["reflect", "zero", "constant", "symmetric", "periodic"]


def dev():
    return torch.device("cuda:0")


def traced(fn):
    _engine.level_events = []
    try:
        out = fn()
        torch.cuda.synchronize()
        kids = [e[1] for e in _engine.level_events]
    finally:
        _engine.level_events = None
    return out, kids


def check(x, wavelet, mode, level, first_kid=_engine.KID_LONG):
    got, kids = traced(lambda: ptwt_amd.wavedec(x.to(dev()), wavelet, mode=mode, level=level))
    want = O.wavedec(x.numpy().astype(np.float64), wavelet, mode=mode, level=level)
    if first_kid is not None:
        assert kids[0] == first_kid, (wavelet, mode, tuple(x.shape), kids)
    assert len(got) == len(want)
    for i, (a, b) in enumerate(zip(got, want)):
        assert tuple(a.shape) == tuple(b.shape), (i, a.shape, b.shape)
        err = G.relerr(a.cpu().numpy(), b)
        assert err < TOL32, f"{wavelet} {mode} L{level} {tuple(x.shape)} coefficient {i}: rel err {err:.3e}"
    return kids


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("wavelet", ["haar", "db2", "db5", "db8", "sym10"])
def test_long_rows_vs_oracle(wavelet, mode):
    """Several interior chunks per row, odd and even lengths at every level, both end pieces; the tail kernel finishes."""
    g = torch.Generator().manual_seed(7)
    for shape, level in (((3, 100003), 8), ((2, 65536), 6), ((1, 40001), 5)):
        x = torch.randn(*shape, generator=g, dtype=torch.float32)
        kids = check(x, wavelet, mode, level)
        assert len(kids) <= 3, kids  # chunked launch (+ a second one for short filters' deep halos) + tail


@pytest.mark.parametrize("mode", MODES)
def test_long_rows_two_levels_and_strided_rows(mode):
    """Rows just beyond twice the one-workgroup limit (two fused levels, few chunks) and rows of a wider tensor whose starts
    are not 16-byte aligned (scalar loads)."""
    g = torch.Generator().manual_seed(8)
    x = torch.randn(2, 33001, generator=g, dtype=torch.float32)
    check(x, "db3", mode, 4)
    wide = torch.randn(4, 70001, generator=g, dtype=torch.float32)
    view = wide[::2, 3:66002]
    xd = wide.to(dev())[::2, 3:66002]
    got, kids = traced(lambda: ptwt_amd.wavedec(xd, "db4", mode=mode, level=7))
    assert kids[0] == _engine.KID_LONG, kids
    want = O.wavedec(view.numpy().astype(np.float64), "db4", mode=mode, level=7)
    for a, b in zip(got, want):
        assert G.relerr(a.cpu().numpy(), b) < TOL32


def test_long_rows_match_per_level_kernels_and_round_trip():
    g = torch.Generator().manual_seed(9)
    x = torch.randn(4, 250000, generator=g, dtype=torch.float32).to(dev())
    for wavelet in ("db5", "sym4"):
        got, kids = traced(lambda: ptwt_amd.wavedec(x, wavelet, mode="periodic", level=10))
        assert kids[0] == _engine.KID_LONG and len(kids) <= 3, kids  # long rows, the same kernel on the shorter rows, (tail)
        _engine.set_option(_engine.OPT_PAIR_MODE, 2)
        try:
            single = ptwt_amd.wavedec(x, wavelet, mode="periodic", level=10)
        finally:
            _engine.set_option(_engine.OPT_PAIR_MODE, 0)
        for a, b in zip(got, single):
            assert a.shape == b.shape and float((a - b).norm() / b.norm()) < 2e-6
        rec = ptwt_amd.waverec(ptwt_amd.wavedec(x, wavelet, level=10), wavelet)
        assert float((rec[..., : x.shape[-1]] - x).abs().max()) < 2e-5


def test_reference_speed_test_shape():
    """The reference's own 1-D speed test (examples/speed_tests/timeitconv_1d.py:10-12): 32 x 10^6 fp32, db5, level 10,
    periodic — two launches; three rows against the oracle."""
    g = torch.Generator(device=dev()).manual_seed(10)
    x = torch.randn(32, 1000000, device=dev(), generator=g)
    got, kids = traced(lambda: ptwt_amd.wavedec(x, "db5", mode="periodic", level=10))
    assert kids == [_engine.KID_LONG, _engine.KID_LONG], kids  # 6 levels of the 10^6-sample rows, 4 of the 15 633-sample rows
    rows = [0, 13, 31]
    want = O.wavedec(x[rows].cpu().numpy().astype(np.float64), "db5", mode="periodic", level=10)
    for a, b in zip(got, want):
        assert G.relerr(a[rows].cpu().numpy(), b) < TOL32


# ---- the finest synthesis levels in one launch (mifwt_dwt1_inv_long, kernel id 18) ---------------------------------------------
@pytest.mark.parametrize("wavelet", ["haar", "db2", "db5", "db8", "sym10"])
def test_long_rows_synthesis_vs_oracle(wavelet):
    """waverec of long rows: coarse levels in the one-workgroup-per-row launch, the fine ones in the chunked launch — against the
    fp64 oracle on the same coefficients (odd and even lengths at every level: with and without the reference's end-crop), and
    against the per-level kernels."""
    rng = np.random.default_rng(17)
    for shape, level in (((3, 100003), 8), ((2, 65536), 6), ((1, 40001), 5), ((2, 1000000), 10)):
        x = rng.standard_normal(shape)
        want_c = O.wavedec(x, wavelet, level=level)
        want = O.waverec(want_c, wavelet)
        cg = [torch.from_numpy(c).float().to(dev()) for c in want_c]
        got, kids = traced(lambda: ptwt_amd.waverec(cg, wavelet))
        assert kids[-1] == _engine.KID_INV_LONG, (wavelet, shape, kids)
        assert got.shape == want.shape and G.relerr(got.cpu().numpy(), want) < 2e-6, (wavelet, shape)
        _engine.set_option(_engine.OPT_PAIR_MODE, 2)
        try:
            single = ptwt_amd.waverec(cg, wavelet)
        finally:
            _engine.set_option(_engine.OPT_PAIR_MODE, 0)
        assert float((got - single).norm() / single.norm()) < 2e-6


def test_long_rows_synthesis_round_trip_and_strided_coefficients():
    g = torch.Generator(device=dev()).manual_seed(19)
    x = torch.randn(32, 1000000, device=dev(), generator=g)
    c = ptwt_amd.wavedec(x, "db5", mode="periodic", level=10)
    rec, kids = traced(lambda: ptwt_amd.waverec(c, "db5"))
    assert kids[-1] == _engine.KID_INV_LONG and len(kids) <= 3, kids
    assert rec.shape[-1] >= x.shape[-1] and float((rec[:, : x.shape[-1]] - x).abs().max()) < 2e-5
    # coefficient rows of wider tensors (row strides that are not the lengths, odd element offsets)
    wide = [torch.randn(t.shape[0], t.shape[1] + 5, device=dev()) for t in c]
    views = [w[:, 3: 3 + t.shape[1]].copy_(t) for w, t in zip(wide, c)]
    rec2 = ptwt_amd.waverec(views, "db5")
    assert torch.equal(rec2, rec)


def test_long_rows_randomised_against_per_level_kernels():
    """Random lengths around the planner's thresholds, filters, modes, row counts and level counts: the chunked launches (both
    directions, long rows and few medium rows) against the per-level kernels on the same data (those are pinned against the
    oracle and the goldens); 2e-6 norm-wise per coefficient vector."""
    rng = np.random.default_rng(2024)
    wavelets = ["haar", "db2", "db3", "db4", "db5", "db7", "sym8", "db10", "coif1", "bior2.2"]
    lengths = [4096, 4097, 5003, 8191, 16384, 16385, 16391, 20000, 32768, 32771, 50001, 65537, 131073, 262145, 300007]
    ran_fwd = ran_inv = 0
    for trial in range(48):
        wavelet = wavelets[rng.integers(len(wavelets))]
        n = lengths[rng.integers(len(lengths))] + int(rng.integers(0, 3))
        rows = int(rng.choice([1, 2, 3, 7, 33, 130]))
        if rows * n > 6_000_000:
            rows = max(1, 6_000_000 // n)
        mode = MODES[rng.integers(len(MODES))]
        flen = len(ptwt_amd._wavelets.host_taps(wavelet)[0])
        maxlev = int(np.floor(np.log2(n / (flen - 1)))) if n >= flen - 1 else 0
        level = int(rng.integers(2, max(3, min(maxlev, 12) + 1)))
        x = torch.randn(rows, n, device=dev())
        try:
            got, kids = traced(lambda: ptwt_amd.wavedec(x, wavelet, mode=mode, level=level))
        except RuntimeError:  # torch's refusal of reflect / circular pads longer than the row, reproduced by the host layer
            continue
        _engine.set_option(_engine.OPT_PAIR_MODE, 2)
        try:
            want = ptwt_amd.wavedec(x, wavelet, mode=mode, level=level)
            rec_want = ptwt_amd.waverec(want, wavelet)
        finally:
            _engine.set_option(_engine.OPT_PAIR_MODE, 0)
        for i, (a, b) in enumerate(zip(got, want)):
            assert a.shape == b.shape, (trial, wavelet, n, rows, mode, level, i)
            err = float((a - b).norm() / b.norm().clamp_min(1e-30))
            assert err < 2e-6, (trial, wavelet, n, rows, mode, level, i, err, kids)
        rec, rkids = traced(lambda: ptwt_amd.waverec(want, wavelet))
        assert rec.shape == rec_want.shape and float((rec - rec_want).norm() / rec_want.norm()) < 2e-6, (trial, wavelet, n, rows, level, rkids)
        ran_fwd += _engine.KID_LONG in kids
        ran_inv += _engine.KID_INV_LONG in rkids
    assert ran_fwd >= 20 and ran_inv >= 20, (ran_fwd, ran_inv)


@pytest.mark.parametrize("mode", ["reflect", "periodic", "zero"])
def test_many_medium_rows_through_the_chunked_launches(mode):
    """Many rows of 1 K .. 16 K samples (the chunked launches serve them too: the two end pieces may be the whole row): against
    the per-level kernels and, on a few rows, the oracle."""
    g = torch.Generator(device=dev()).manual_seed(31)
    for shape, wavelet, level in [((600, 4099), "db4", 6), ((300, 16384), "db5", 8), ((2000, 1031), "db2", 5), ((500, 8192), "haar", 10),
                                  ((257, 12001), "sym8", 7)]:
        x = torch.randn(*shape, device=dev(), generator=g)
        got, kids = traced(lambda: ptwt_amd.wavedec(x, wavelet, mode=mode, level=level))
        _engine.set_option(_engine.OPT_PAIR_MODE, 2)
        try:
            want = ptwt_amd.wavedec(x, wavelet, mode=mode, level=level)
            rec_want = ptwt_amd.waverec(want, wavelet)
        finally:
            _engine.set_option(_engine.OPT_PAIR_MODE, 0)
        if shape[1] >= 4096:
            assert kids[0] == _engine.KID_LONG, (shape, kids)
        for i, (a, b) in enumerate(zip(got, want)):
            assert a.shape == b.shape and float((a - b).norm() / b.norm()) < 2e-6, (shape, wavelet, mode, i, kids)
        rows = [0, shape[0] // 2, shape[0] - 1]
        ref = O.wavedec(x[rows].cpu().numpy().astype(np.float64), wavelet, mode=mode, level=level)
        for a, b in zip(got, ref):
            assert G.relerr(a[rows].cpu().numpy(), b) < TOL32
        rec, rkids = traced(lambda: ptwt_amd.waverec(want, wavelet))
        assert _engine.KID_INV_LONG in rkids, (shape, rkids)
        assert rec.shape == rec_want.shape and float((rec - rec_want).norm() / rec_want.norm()) < 2e-6, (shape, wavelet, rkids)


@pytest.mark.parametrize("wavelet", ["haar", "db3", "db5", "sym10"])
def test_long_rows_synthesis_f64(wavelet):
    """The chunked synthesis launch on f64 data (the reference's second dtype): against the fp64 oracle at 1e-12 and bit for bit
    against the per-level kernels (same summation order)."""
    rng = np.random.default_rng(23)
    for shape, level in (((3, 100003), 8), ((2, 65536), 6), ((40, 9001), 5), ((300, 4099), 4)):
        x = rng.standard_normal(shape)
        want_c = O.wavedec(x, wavelet, level=level)
        want = O.waverec(want_c, wavelet)
        cg = [torch.from_numpy(c).to(dev()) for c in want_c]
        got, kids = traced(lambda: ptwt_amd.waverec(cg, wavelet))
        assert kids[-1] == _engine.KID_INV_LONG, (wavelet, shape, kids)
        assert got.dtype == torch.float64 and got.shape == want.shape and G.relerr(got.cpu().numpy(), want) < 1e-12, (wavelet, shape)
        _engine.set_option(_engine.OPT_PAIR_MODE, 2)
        try:
            single = ptwt_amd.waverec(cg, wavelet)
        finally:
            _engine.set_option(_engine.OPT_PAIR_MODE, 0)
        assert float((got - single).abs().max()) < 1e-13 * float(single.abs().max())
