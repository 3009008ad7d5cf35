"""Pin the CPU oracle (oracle/fwt_oracle.py) against the reference's own ground truth.

(1) the reference's known-answer test, (2) real-PyWavelets goldens, (3) goldens produced by the reference
itself, (4) the live reference when /root/reference is importable (build container only).
"""
import os
import sys

import numpy as np
import pytest

from oracle import fwt_oracle as O
from tests import _golden as G

TOL64 = 1e-12  # fp64 norm-wise tolerance vs reference/pywt
TOL32 = 1e-6   # fp32 norm-wise tolerance per sub-band


def test_filter_bank_table_complete():
    import json

    with open(os.path.join(G.GOLDEN, "pywt_filter_banks.json")) as f:
        banks = json.load(f)
    names = [k for k in banks if not k.startswith("_")]
    assert len(names) == 106
    assert len(banks["coif17"]["dec_lo"]) == 102
    for n in names:
        fb = banks[n]
        assert len(fb["dec_lo"]) == len(fb["dec_hi"]) == len(fb["rec_lo"]) == len(fb["rec_hi"])
        assert len(fb["dec_lo"]) % 2 == 0


def test_kat_ripples_haar_lvl3():
    """Reference tests/test_convolution_fwt.py:98-118 ("Ripples in Mathematics", p. 7)."""
    bank = ([0.5, 0.5], [-0.5, 0.5], [0.5, 0.5], [0.5, -0.5])
    data = np.array([56.0, 40.0, 8.0, 24.0, 48.0, 48.0, 40.0, 16.0])
    c = O.wavedec(data, bank, level=3)
    assert c[0].item() == 35.0
    assert c[1].item() == -3.0
    assert (c[2] == [16.0, 10.0]).all()
    assert (c[3] == [8.0, -8.0, 0.0, 12.0]).all()


@pytest.mark.parametrize("case", G.pywt1d_cases(), ids=lambda c: "%s-%d-%s-L%d" % (c["wavelet"], c["n"], c["mode"], c["level"]))
def test_oracle_vs_pywt_1d(case):
    z, _ = G.load("pywt_wavedec1d.npz")
    x = z[case["key"] + "_x"]
    got = O.wavedec(x, case["wavelet"], mode=case["mode"], level=case["level"])
    assert len(got) == case["ncoef"]
    for i, g in enumerate(got):
        assert G.relerr(g, z["%s_%d" % (case["key"], i)]) < TOL64
    rec = O.waverec(got, case["wavelet"])
    assert G.relerr(rec[..., : x.shape[-1]], x) < 1e-9  # bior/rbio taps are only ~1e-11 PR-exact


@pytest.mark.parametrize("case", G.pywt2d_cases(), ids=lambda c: "%s-%s-%s-L%d" % (c["wavelet"], "x".join(map(str, c["shape"])), c["mode"], c["level"]))
def test_oracle_vs_pywt_2d(case):
    z, _ = G.load("pywt_wavedec2d.npz")
    k = case["key"]
    x = z[k + "_x"]
    got = O.wavedec2(x, case["wavelet"], mode=case["mode"], level=case["level"])
    assert G.relerr(got[0], z[k + "_a"]) < TOL64
    for i, (h, v, d) in enumerate(got[1:]):
        assert G.relerr(h, z["%s_%d_h" % (k, i)]) < TOL64
        assert G.relerr(v, z["%s_%d_v" % (k, i)]) < TOL64
        assert G.relerr(d, z["%s_%d_d" % (k, i)]) < TOL64
    rec = O.waverec2(got, case["wavelet"])
    assert G.relerr(rec[..., : x.shape[-2], : x.shape[-1]], x) < 1e-9


@pytest.mark.parametrize("case", G.pywt3d_cases(), ids=lambda c: "%s-%s-%s-L%d" % (c["wavelet"], "x".join(map(str, c["shape"])), c["mode"], c["level"]))
def test_oracle_vs_pywt_3d(case):
    z, _ = G.load("pywt_wavedec3d.npz")
    k = case["key"]
    x = z[k + "_x"]
    got = O.wavedec3(x, case["wavelet"], mode=case["mode"], level=case["level"])
    assert G.relerr(got[0], z[k + "_a"]) < TOL64
    for i, dct in enumerate(got[1:]):
        assert list(dct.keys()) == ["aad", "ada", "add", "daa", "dad", "dda", "ddd"]
        for key, val in dct.items():
            assert G.relerr(val, z["%s_%d_%s" % (k, i, key)]) < TOL64
    rec = O.waverec3(got, case["wavelet"])
    s = x.shape
    assert G.relerr(rec[..., : s[-3], : s[-2], : s[-1]], x) < 1e-9


def _run_oracle(case, x):
    kw = dict(case["kw"])
    for key in ("axes",):
        if key in kw:
            kw[key] = tuple(kw[key])
    coeffs = getattr(O, case["fn"])(x, case["wavelet"], **kw)
    rkw = {k: v for k, v in kw.items() if k in ("axis", "axes")}
    rec = getattr(O, case["rec"])(coeffs, case["wavelet"], **rkw)
    return coeffs, rec


@pytest.mark.parametrize("case", G.ref_cases(), ids=lambda c: "%s-%s-%s-%s" % (c["key"], c["fn"], c["wavelet"], c["dtype"]))
def test_oracle_vs_reference_goldens(case):
    z, _ = G.load("ptwt_ref.npz")
    k = case["key"]
    x = z[k + "_x"]
    tol = TOL64 if case["dtype"] == "float64" else TOL32
    coeffs, rec = _run_oracle(case, x)
    flat = G.flatten_coeffs(coeffs)
    assert [n for n, _ in flat] == case["names"]  # container structure and key order
    for name, val in flat:
        want = z["%s_%s" % (k, name)]
        assert val.dtype == want.dtype
        assert G.relerr(val, want) < tol, name
    want = z[k + "_rec"]
    assert rec.shape == want.shape  # includes the "+1 sample for odd extents" behaviour
    assert G.relerr(rec, want) < (1e-11 if case["dtype"] == "float64" else 2e-6)


def test_baseline_config1_haar_4096_roundtrip():
    """BASELINE.json configs[0]: 1-D Haar, N=4096, batch 1, fp64, 12 levels, round trip."""
    rng = np.random.default_rng(7)
    x = rng.standard_normal((1, 4096))
    c = O.wavedec(x, "haar")
    assert [t.shape[-1] for t in c] == [1, 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048]
    assert np.abs(O.waverec(c, "haar") - x).max() < 1e-13


def test_oracle_error_behaviour():
    x = np.zeros((4, 8))
    with pytest.raises(ValueError):
        O.wavedec(x, "haar", mode="nope")
    with pytest.raises(ValueError):
        O.wavedec2(np.zeros(8), "haar")
    with pytest.raises(ValueError):
        O.wavedec(x.astype(np.float16), "haar")
    with pytest.raises(RuntimeError):
        O.wavedec(np.zeros((1, 6)), "db4", mode="reflect", level=1)  # pad 6 >= N 6
    # symmetric tolerates pad > N
    O.wavedec(np.zeros((1, 4)), "db4", mode="symmetric", level=1)


REF = "/root/reference/src"


@pytest.mark.skipif(not os.path.isdir(REF), reason="live reference only exists in the build container")
def test_oracle_vs_live_reference():
    stubs = os.path.join(G.GOLDEN, "_stubs")
    sys.path[:0] = [stubs, REF]
    try:
        import torch

        import ptwt

        rng = np.random.default_rng(11)
        x = rng.standard_normal((2, 37, 41))
        for mode in O.MODES:
            ref = ptwt.wavedec2(torch.from_numpy(x), "db3", mode=mode, level=2)
            got = O.wavedec2(x, "db3", mode=mode, level=2)
            for (n, a), (_, b) in zip(G.flatten_coeffs(got), G.flatten_coeffs(ref)):
                assert G.relerr(a, b.numpy()) < TOL64, (mode, n)
            assert G.relerr(O.waverec2(got, "db3"), ptwt.waverec2(ref, "db3").numpy()) < 1e-11
        x3 = rng.standard_normal((9, 10, 11))
        ref = ptwt.fswavedec3(torch.from_numpy(x3), "db2", level=1)
        got = O.fswavedec3(x3, "db2", level=1)
        assert list(ref[1].keys()) == list(got[1].keys())
        for key in ref[1]:
            assert G.relerr(got[1][key], ref[1][key].numpy()) < TOL64
    finally:
        del sys.path[:2]
        for m in [m for m in sys.modules if m == "pywt" or m.startswith(("pywt.", "ptwt")) or m == "more_itertools"]:
            del sys.modules[m]


@pytest.mark.parametrize("mode", O.MODES)
def test_torch_cpu_port_matches_oracle(mode):
    """The cpu_baseline port (dense conv2d, as the reference does on CPU) equals the separable oracle."""
    import torch

    from oracle import torch_cpu_port as P

    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 45, 52))
    want = O.wavedec2(x, "db4", mode=mode, level=2)
    got = P.wavedec2(torch.from_numpy(x), "db4", mode=mode, level=2)
    for (n, a), (_, b) in zip(G.flatten_coeffs(got), G.flatten_coeffs(want)):
        assert G.relerr(a.numpy(), b) < TOL64, (mode, n)


@pytest.mark.parametrize("mode", O.MODES)
def test_torch_cpu_ports_of_the_other_functions_match_oracle(mode):
    """The cpu_baseline ports of wavedec / wavedec3 / fswavedec2 / waverec2 / fswaverec2 / waverec / waverec3 (the reference's ATen op
    sequences) equal the oracle."""
    import torch

    from oracle import torch_cpu_port as P

    rng = np.random.default_rng(6)
    x1 = rng.standard_normal((3, 101))
    for a, b in zip(P.wavedec(torch.from_numpy(x1), "db3", mode=mode, level=3), O.wavedec(x1, "db3", mode=mode, level=3)):
        assert G.relerr(a.numpy(), b) < TOL64, mode
    x3 = rng.standard_normal((2, 20, 21, 22))
    got, want = P.wavedec3(torch.from_numpy(x3), "db2", mode=mode, level=2), O.wavedec3(x3, "db2", mode=mode, level=2)
    assert G.relerr(got[0].numpy(), want[0]) < TOL64
    for g, w in zip(got[1:], want[1:]):
        assert list(g) == list(w)
        for k in w:
            assert G.relerr(g[k].numpy(), w[k]) < TOL64, (mode, k)
    x2 = rng.standard_normal((2, 37, 50))
    got, want = P.fswavedec2(torch.from_numpy(x2), "db3", mode=mode, level=2), O.fswavedec2(x2, "db3", mode=mode, level=2)
    assert G.relerr(got[0].numpy(), want[0]) < TOL64
    for g, w in zip(got[1:], want[1:]):
        for k in w:
            assert G.relerr(g[k].numpy(), w[k]) < TOL64, (mode, k)
    for wav in ("haar", "db3", "db4"):
        c = O.wavedec2(x2, wav, mode=mode, level=3)
        rec = P.waverec2(tuple([torch.from_numpy(c[0])] + [tuple(torch.from_numpy(v) for v in lv) for lv in c[1:]]), wav)
        assert G.relerr(rec.numpy(), O.waverec2(c, wav)) < TOL64, (mode, wav)
        c = O.fswavedec2(x2, wav, mode=mode, level=3)
        rec = P.fswaverec2(tuple([torch.from_numpy(c[0])] + [{k: torch.from_numpy(v) for k, v in d.items()} for d in c[1:]]), wav)
        assert G.relerr(rec.numpy(), O.fswaverec2(c, wav)) < TOL64, (mode, wav)
        c = O.wavedec(x1, wav, mode=mode, level=3)
        assert G.relerr(P.waverec([torch.from_numpy(t) for t in c], wav).numpy(), O.waverec(c, wav)) < TOL64, (mode, wav)
        c = O.wavedec3(x3, wav, mode=mode, level=2)
        rec = P.waverec3(tuple([torch.from_numpy(c[0])] + [{k: torch.from_numpy(v) for k, v in d.items()} for d in c[1:]]), wav)
        assert G.relerr(rec.numpy(), O.waverec3(c, wav)) < TOL64, (mode, wav)
