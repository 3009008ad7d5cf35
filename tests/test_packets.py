"""Wavelet packet trees (``ptwt_amd.WaveletPacket`` / ``WaveletPacket2D``) against golden trees of the reference's
own packet classes (tests/golden/ptwt_ref_packets.npz, made by make_ptwt_ref_packet_goldens.py).

CPU tests run the tree logic on the oracle level engine (tests/_oracle_engine.py); the ``gpu`` tests run it on the
HIP engine, where a packet level is ONE launch over all nodes."""
import numpy as np
import pytest
import torch

import ptwt_amd
from ptwt_amd import _engine
from tests import _golden as G
from tests._oracle_engine import OracleLevelEngine


@pytest.fixture()
def oracle_engine(monkeypatch):
    monkeypatch.setattr(_engine, "ENGINE", OracleLevelEngine())


def _run_cases(device, tol):
    z, idx = G.load("ptwt_ref_packets.npz")
    for case in idx:
        k = case["key"]
        kw = {a: (tuple(v) if isinstance(v, list) else v) for a, v in case["kw"].items()}
        x = torch.from_numpy(z[k + "_x"]).to(device)
        cls = ptwt_amd.WaveletPacket if case["dim"] == 1 else ptwt_amd.WaveletPacket2D
        wp = cls(x, case["wavelet"], mode=case["mode"], maxlevel=case["maxlevel"], **kw)
        assert wp.get_level(case["maxlevel"], "natural") == case["keys"]
        for key in case["keys"]:
            want = z["%s_n_%s" % (k, key)]
            got = wp[key]
            assert tuple(got.shape) == want.shape, (case, key)
            assert G.relerr(got.cpu().numpy(), want) < tol, (case, key)
        for key in case["keys"]:
            wp[key] = 0.5 * wp[key]
        wp.reconstruct()
        want = z[k + "_rec"]
        assert tuple(wp[""].shape) == want.shape, case
        assert G.relerr(wp[""].cpu().numpy(), want) < tol, (case, "reconstruct")


def test_packets_vs_reference_goldens_cpu(oracle_engine):
    _run_cases(torch.device("cpu"), 1e-12)


def test_node_orderings():
    WP, WP2 = ptwt_amd.WaveletPacket, ptwt_amd.WaveletPacket2D
    assert WP.get_level(0) == [""] and WP.get_level(1) == ["a", "d"]
    assert WP.get_level(2) == ["aa", "ad", "dd", "da"]  # Gray code
    assert WP.get_level(3, "freq") == ["aaa", "aad", "add", "ada", "dda", "ddd", "dad", "daa"]
    assert WP.get_level(2, "natural") == ["aa", "ad", "da", "dd"]
    assert WP2.get_natural_order(1) == ["a", "h", "v", "d"]
    assert WP2.get_freq_order(1) == [["a", "v"], ["h", "d"]]
    f2 = WP2.get_freq_order(2)
    assert len(f2) == 4 and all(len(r) == 4 for r in f2) and f2[0][0] == "aa"
    assert sorted(sum(f2, [])) == sorted(WP2.get_natural_order(2))
    with pytest.raises(ValueError):
        WP.get_level(2, "random")
    with pytest.raises(ValueError):
        WP2.get_level(2, "random")


def test_lazy_access_and_errors(oracle_engine):
    x = torch.randn(2, 64, dtype=torch.float64)
    wp = ptwt_amd.WaveletPacket(None, "db2")
    with pytest.raises(ValueError):
        wp["a"]  # not initialised
    wp.transform(x, maxlevel=2)
    assert list(wp.keys()) == [""]
    wp["ad"]  # expands levels 1 and 2 (whole levels: one launch each)
    assert set(wp.keys()) == {"", "a", "d", "aa", "ad", "da", "dd"}
    with pytest.raises(KeyError):
        wp["aaa"]  # deeper than maxlevel
    with pytest.raises(ValueError):
        wp["ax"]  # invalid char
    with pytest.raises(NotImplementedError):
        ptwt_amd.WaveletPacket(x, "db2", mode="boundary")
    # a node assigned before its children exist feeds their expansion
    wp = ptwt_amd.WaveletPacket(x, "haar", maxlevel=2)
    wp["a"] = torch.zeros_like(wp["a"])
    assert float(wp["aa"].abs().max()) == 0.0 and float(wp["da"].abs().max()) > 0.0
    # reconstruct needs every leaf
    wp = ptwt_amd.WaveletPacket(x, "haar", maxlevel=2)
    wp["aa"]
    del wp.data["dd"]
    with pytest.raises(KeyError):
        wp.reconstruct()
    # default maxlevel = dwt_max_level
    assert ptwt_amd.WaveletPacket(x, "db2").maxlevel == 4
    assert ptwt_amd.WaveletPacket2D(torch.randn(20, 33, dtype=torch.float64), "db2").maxlevel == 2


@pytest.mark.gpu
def test_packets_vs_reference_goldens_gpu():
    _run_cases(torch.device("cuda:0"), 1e-12)


@pytest.mark.gpu
def test_packet_level_is_one_launch_fp32():
    """Full level-3 2-D tree of a batch of images: 3 launches (one per level), every node checked against level-1
    wavedec2 calls on its parent; round trip through reconstruct()."""
    dev = torch.device("cuda:0")
    x = torch.randn(8, 256, 256, device=dev)
    _engine.level_events = []
    try:
        wp = ptwt_amd.WaveletPacket2D(x, "db4", mode="symmetric", maxlevel=3)
        nodes = [wp[k] for k in wp.get_level(3, "natural")]
        torch.cuda.synchronize()
        assert len(_engine.level_events) == 3
    finally:
        _engine.level_events = None
    assert len(nodes) == 64
    a, (h, v, d) = ptwt_amd.wavedec2(wp["hv"], "db4", mode="symmetric", level=1)
    for key, want in (("hva", a), ("hvh", h), ("hvv", v), ("hvd", d)):
        assert G.relerr(wp[key].cpu().numpy(), want.cpu().numpy()) < 1e-6, key
    wp.reconstruct()
    assert G.relerr(wp[""].cpu().numpy(), x.cpu().numpy()) < 2e-6
