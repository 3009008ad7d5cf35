"""Generate golden vectors from the REAL PyWavelets (third-party dependency of the reference).

Run with the interpreter that has PyWavelets (this image: /opt/conda/bin/python3.9, PyWavelets 1.1.1):

    /opt/conda/bin/python3.9 tests/golden/make_pywt_goldens.py

The reference (ptwt) takes its filter taps and max-level formula from PyWavelets
(reference call sites: src/ptwt/_util.py:82,121; conv_transform.py:131; conv_transform_2.py:138;
conv_transform_3.py:117-119) and every hot-path test of the reference pins ptwt against live
``pywt.wavedec / wavedec2 / wavedecn`` output (reference tests/test_convolution_fwt.py:21-61,170-267;
tests/test_convolution_fwt_3.py:51-97).  These files freeze those outputs so they can travel to a box
without PyWavelets.

Outputs (all under tests/golden/):
  pywt_filter_banks.json   taps of all discrete wavelets (dec_lo, dec_hi, rec_lo, rec_hi)
  pywt_wavedec1d.npz       pywt.wavedec  cases
  pywt_wavedec2d.npz       pywt.wavedec2 cases
  pywt_wavedec3d.npz       pywt.wavedecn cases (axes=-3..-1)
Inputs are stored alongside the outputs (numpy's Generator stream is stable, but we do not rely on it).
"""
import json
import os
import warnings

import numpy as np
import pywt

HERE = os.path.dirname(os.path.abspath(__file__))
MODES = ["reflect", "zero", "constant", "periodic", "symmetric"]
warnings.simplefilter("ignore")


def ref_pad_ok(n, filt_len, mode):
    """torch raises for reflect pad >= N and circular pad > N; ptwt surfaces that error."""
    pad = filt_len - 2 + (n % 2)
    if mode == "reflect":
        return pad < n
    if mode == "periodic":
        return pad <= n
    return True


def levels_ok(shape, filt_len, mode, level):
    """Check every level of the pyramid is paddable by the reference."""
    cur = list(shape)
    for _ in range(level):
        if not all(ref_pad_ok(n, filt_len, mode) for n in cur):
            return False
        cur = [(n + filt_len - 1) // 2 for n in cur]
    return True


def dump_filter_banks():
    out = {"_pywt_version": pywt.__version__}
    for name in pywt.wavelist(kind="discrete"):
        w = pywt.Wavelet(name)
        out[name] = {
            "dec_lo": list(map(float, w.dec_lo)),
            "dec_hi": list(map(float, w.dec_hi)),
            "rec_lo": list(map(float, w.rec_lo)),
            "rec_hi": list(map(float, w.rec_hi)),
        }
    with open(os.path.join(HERE, "pywt_filter_banks.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("filter banks:", len(out) - 1)


def dump_1d():
    rng = np.random.default_rng(0)
    store, index = {}, []
    wavelets = ["haar", "db2", "db4", "db5", "db8", "sym5", "sym16", "bior2.2", "rbio2.4", "coif3"]
    for wname in wavelets:
        w = pywt.Wavelet(wname)
        for n in (64, 65, 37):
            x = rng.standard_normal((2, n))
            for mode in MODES:
                for level in (1, 2, None):
                    lv = pywt.dwt_max_level(n, w.dec_len) if level is None else level
                    if lv < 1 or not levels_ok([n], w.dec_len, mode, lv):
                        continue
                    cs = pywt.wavedec(x, w, mode=mode, level=lv, axis=-1)
                    key = "c%04d" % len(index)
                    index.append(dict(key=key, wavelet=wname, n=n, mode=mode, level=lv, ncoef=len(cs)))
                    store[key + "_x"] = x
                    for i, c in enumerate(cs):
                        store["%s_%d" % (key, i)] = c
    store["index"] = np.array(json.dumps(index))
    np.savez_compressed(os.path.join(HERE, "pywt_wavedec1d.npz"), **store)
    print("1d cases:", len(index))


def dump_2d():
    rng = np.random.default_rng(1)
    store, index = {}, []
    for wname in ["haar", "db2", "db4", "sym5", "bior2.2", "rbio2.4"]:
        w = pywt.Wavelet(wname)
        for shape in ((31, 33), (24, 40)):
            x = rng.standard_normal((1,) + shape)
            for mode in MODES:
                for level in (1, 2, None):
                    lv = pywt.dwtn_max_level(shape, w) if level is None else level
                    if lv < 1 or not levels_ok(shape, w.dec_len, mode, lv):
                        continue
                    if level is None and lv in (1, 2):
                        continue  # duplicate of an explicit level
                    cs = pywt.wavedec2(x, w, mode=mode, level=lv, axes=(-2, -1))
                    key = "c%04d" % len(index)
                    index.append(dict(key=key, wavelet=wname, shape=list(shape), mode=mode, level=lv))
                    store[key + "_x"] = x
                    store[key + "_a"] = cs[0]
                    for i, (h, v, d) in enumerate(cs[1:]):
                        store["%s_%d_h" % (key, i)] = h
                        store["%s_%d_v" % (key, i)] = v
                        store["%s_%d_d" % (key, i)] = d
    store["index"] = np.array(json.dumps(index))
    np.savez_compressed(os.path.join(HERE, "pywt_wavedec2d.npz"), **store)
    print("2d cases:", len(index))


def dump_3d():
    rng = np.random.default_rng(2)
    store, index = {}, []
    for wname in ["haar", "db2", "db4"]:
        w = pywt.Wavelet(wname)
        for shape in ((10, 10, 10), (9, 10, 11)):
            x = rng.standard_normal((1,) + shape)
            for mode in MODES:
                for level in (1, 2):
                    if not levels_ok(shape, w.dec_len, mode, level):
                        continue
                    cs = pywt.wavedecn(x, w, mode=mode, level=level, axes=(-3, -2, -1))
                    key = "c%04d" % len(index)
                    index.append(dict(key=key, wavelet=wname, shape=list(shape), mode=mode, level=level))
                    store[key + "_x"] = x
                    store[key + "_a"] = cs[0]
                    for i, dct in enumerate(cs[1:]):
                        for k, v in dct.items():
                            store["%s_%d_%s" % (key, i, k)] = v
    store["index"] = np.array(json.dumps(index))
    np.savez_compressed(os.path.join(HERE, "pywt_wavedec3d.npz"), **store)
    print("3d cases:", len(index))


if __name__ == "__main__":
    dump_filter_banks()
    dump_1d()
    dump_2d()
    dump_3d()
