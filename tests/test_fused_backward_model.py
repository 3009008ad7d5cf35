"""The algebra behind the one-launch backward of `waverec2` / `waverec` (`_fwt._SynthesisPyramid.backward`, `_SynthesisChain1d.backward`,
round 5), checked with the oracle alone: the transpose of a whole reconstruction (src/ptwt/conv_transform_2.py:222-249 per level, with
the reference's end-crop of a level's output where the next detail band is one sample shorter, src/ptwt/_util.py:231-244) IS a
zero-mode multi-level ANALYSIS with the reconstruction taps reversed — level by level the coefficient shapes agree, and a cropped
sample is a zero the zero extension supplies anyway.  So <waverec(c), g> = <c, wavedec_zero_reversed(g)> band by band; the GPU tests
compare the launches themselves with the per-level adjoints."""
import numpy as np
import pytest

from oracle import fwt_oracle as O


def _reversed_bank(wavelet):
    dec_lo, dec_hi, rec_lo, rec_hi = O.filter_bank(wavelet)
    return (rec_lo[::-1].copy(), rec_hi[::-1].copy(), rec_lo, rec_hi)  # analysis taps := reversed synthesis taps


@pytest.mark.parametrize("wavelet", ["haar", "db2", "db4", "sym5"])
@pytest.mark.parametrize("mode", ["reflect", "zero", "periodic", "symmetric"])
@pytest.mark.parametrize("shape,level", [((37, 52), 3), ((64, 64), 2), ((45, 41), 2), ((90, 33), 3)])
def test_transpose_of_waverec2_is_zero_mode_wavedec2_with_reversed_rec_taps(wavelet, mode, shape, level):
    rng = np.random.default_rng(len(wavelet) + shape[0])
    flen = len(O.filter_bank(wavelet)[0])
    if min(shape) < (2 ** (level - 1)) * flen:
        pytest.skip("plane too small for this depth")
    try:
        coeffs = O.wavedec2(rng.standard_normal(shape), wavelet, mode=mode, level=level)  # (only its SHAPES matter: odd sizes give trims)
    except RuntimeError:
        pytest.skip("the reference refuses this pad")
    c = (rng.standard_normal(coeffs[0].shape),) + tuple(tuple(rng.standard_normal(b.shape) for b in lev) for lev in coeffs[1:])
    y = O.waverec2(c, wavelet)
    g = rng.standard_normal(y.shape)
    adj = O.wavedec2(g, _reversed_bank(wavelet), mode="zero", level=level)
    assert adj[0].shape == c[0].shape and all(a.shape == b.shape for la, lb in zip(adj[1:], c[1:]) for a, b in zip(la, lb))
    lhs = float((y * g).sum())
    rhs = float((c[0] * adj[0]).sum() + sum((a * b).sum() for la, lb in zip(adj[1:], c[1:]) for a, b in zip(la, lb)))
    assert abs(lhs - rhs) <= 1e-10 * max(1.0, abs(lhs)), (lhs, rhs)
    # ... and band by band: the gradient of one band alone
    for lvl in range(1, level + 1):
        for k in range(3):
            e = tuple([np.zeros_like(c[0])] + [tuple(np.zeros_like(b) for b in lev) for lev in c[1:]])
            e[lvl][k][...] = c[lvl][k]
            assert abs(float((O.waverec2(e, wavelet) * g).sum()) - float((c[lvl][k] * adj[lvl][k]).sum())) <= 1e-10 * max(1.0, abs(lhs))


@pytest.mark.parametrize("wavelet", ["haar", "db3", "db5"])
@pytest.mark.parametrize("n,level", [(1001, 5), (4096, 6), (777, 3)])
def test_transpose_of_waverec_is_zero_mode_wavedec_with_reversed_rec_taps(wavelet, n, level):
    rng = np.random.default_rng(n)
    coeffs = O.wavedec(rng.standard_normal((2, n)), wavelet, mode="symmetric", level=level)
    c = [rng.standard_normal(t.shape) for t in coeffs]
    y = O.waverec(c, wavelet)
    g = rng.standard_normal(y.shape)
    adj = O.wavedec(g, _reversed_bank(wavelet), mode="zero", level=level)
    assert [a.shape for a in adj] == [t.shape for t in c]
    lhs, rhs = float((y * g).sum()), float(sum((a * b).sum() for a, b in zip(adj, c)))
    assert abs(lhs - rhs) <= 1e-10 * max(1.0, abs(lhs)), (lhs, rhs)
