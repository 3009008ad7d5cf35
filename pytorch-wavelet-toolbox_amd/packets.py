"""Wavelet packet trees on the MI355X engine: ``WaveletPacket`` (1-D) and ``WaveletPacket2D``.

API of reference src/ptwt/packets.py:67-360 (1-D) and :362-771 (2-D): dict-like objects whose keys are filter
paths (``"aad"`` / ``"ahv"``), filled lazily on access, with ``transform``, ``initialize``, ``reconstruct`` and
the node-ordering helpers.  The reference expands ONE node per ``wavedec(level=1)`` call (:312-316, :528-539), so
a full level-s tree costs (2^s - 1) resp. (4^s - 1)/3 conv launches.

MI355X-first restatement: a packet level is ONE engine launch.  All nodes of a level live in a single buffer
``[batch, band_1, .., band_s, extents..]`` whose leading dims fold into the kernel's batch — the level-(s+1) buffer
``[batch * nb^s, nb, M..]`` IS that layout for s + 1, no copies — and a node is a strided view of it.  A miss on a
key therefore expands whole levels down to the requested depth (s launches for depth s).  Nodes assigned by the
user are honoured: a level that contains assigned nodes is re-gathered before it is expanded or reconstructed
from.  The sparse-matrix ``mode="boundary"`` backend is out of scope of this engine.
"""
from __future__ import annotations

import collections
from itertools import product
from typing import Dict, Iterable, List, Optional, Sequence, Tuple, Union

import torch

from . import _engine, _fwt
from ._wavelets import as_wavelet, dwt_max_level, filter_length, host_taps

__all__ = ["WaveletPacket", "WaveletPacket2D"]


def _graycode(level: int, lo: str, hi: str) -> List[str]:
    """Frequency (Gray-code) ordering of the 2^level paths over the alphabet (lo, hi)."""
    if level == 0:
        return [""]
    order = [lo, hi]
    for _ in range(level - 1):
        order = [lo + p for p in order] + [hi + p for p in reversed(order)]
    return order


def _wpfreq(fs: float, level: int) -> List[float]:
    """Centre frequencies of a fully decomposed 1-D packet level (src/ptwt/packets.py:49-64)."""
    n = 2 ** level
    return [(fs / 2.0) * (i / n) for i in range(n)]


class _PacketTree(collections.UserDict):
    """Shared machinery: level buffers, lazy whole-level expansion, batched reconstruction."""

    _ndim = 1
    _bands: Dict[str, int] = {}  # key char -> band index of the engine's level buffer

    def __init__(self, data, wavelet, mode, maxlevel, axes) -> None:
        super().__init__()
        self.wavelet = as_wavelet(wavelet)
        self.mode = mode
        if mode == "boundary":
            raise NotImplementedError(
                "mode='boundary' selects the reference's sparse-matrix backend (src/ptwt/matmul_transform*.py), which is "
                "outside this engine's scope; use a padding mode."
            )
        self._axes = _fwt._ensure_axes(axes, self._ndim)
        self._filter_keys = set(self._bands)
        self.maxlevel: Optional[int] = None
        self._layout: Optional[_fwt._Layout] = None
        self._levels: List[Optional[torch.Tensor]] = []  # level s: [B, nb, .. (s times) .., extents..]
        self._assigned: set = set()  # keys whose tensor was set by the user, not produced by this tree
        if data is not None:
            self.transform(data, maxlevel)

    # ---- construction -------------------------------------------------------------------------------------
    def transform(self, data: torch.Tensor, maxlevel: Optional[int] = None):
        """(Re)initialise the tree lazily with ``data`` (src/ptwt/packets.py:149-177, :436-467)."""
        self.data = {}
        self._assigned = set()
        self._layout = _fwt._Layout(data, self._ndim, self._axes)
        root = self._layout.fold(data)
        self._levels = [root]
        self.data[""] = data
        if maxlevel is None:
            maxlevel = dwt_max_level(min(root.shape[1:]), filter_length(self.wavelet))
        self.maxlevel = maxlevel
        return self

    def initialize(self, keys: Iterable[str]) -> None:
        """Compute the nodes in ``keys`` (src/ptwt/packets.py:179-188)."""
        for key in keys:
            self[key]

    # ---- level buffers --------------------------------------------------------------------------------------
    def _band_path(self, key: str) -> Tuple[int, ...]:
        return tuple(self._bands[c] for c in key)

    def _node_view(self, level_buf: torch.Tensor, key: str) -> torch.Tensor:
        idx = (slice(None),) + self._band_path(key)
        return self._layout.unfold(level_buf[idx])

    def _keys_of_level(self, level: int) -> List[str]:
        return ["".join(p) for p in product(sorted(self._bands, key=self._bands.get), repeat=level)]

    def _level_buffer(self, level: int) -> torch.Tensor:
        """The [B, nb^level.., extents..] buffer of ``level``, re-gathered when the user assigned nodes of it."""
        buf = self._levels[level] if level < len(self._levels) else None
        keys = self._keys_of_level(level)
        if buf is not None and not any(k in self._assigned for k in keys):
            return buf
        nodes = []
        for k in keys:
            if k not in self.data:
                if buf is None:
                    raise KeyError(f"Key {k} not found")
                nodes.append(buf[(slice(None),) + self._band_path(k)])
            else:
                nodes.append(self._layout.fold(self.data[k]))
        nb = len(self._bands)
        stacked = torch.stack(nodes, dim=1)  # [B, nb^level (key order = band order), extents..]
        return stacked.reshape(stacked.shape[0], *([nb] * level), *stacked.shape[2:])

    def _expand_level(self, level: int) -> None:
        """Compute every node of ``level + 1`` from ``level`` with ONE analysis launch."""
        src = self._level_buffer(level)
        nb = len(self._bands)
        flat = src.reshape(-1, *src.shape[1 + level:])
        dec_lo, dec_hi, _, _ = host_taps(self.wavelet)
        mode_id = _fwt._mode_id(self.mode)
        _fwt._check_pad(flat.shape[1:], len(dec_lo), "reflect" if self.mode is None else self.mode)
        tap_t = _fwt._tap_tensors(self.wavelet)  # learnable filter bank: the taps stay in the graph (src/ptwt/_util.py:115-132)
        if torch.is_grad_enabled() and (flat.requires_grad or tap_t is not None):
            out = _fwt._AnalysisLevel.apply(flat, dec_lo, dec_hi, mode_id, *((tap_t[0], tap_t[1]) if tap_t else (None, None)))
        else:
            out = _engine.ENGINE.analysis(flat, dec_lo, dec_hi, mode_id)  # [B * nb^level, nb, M..]
        nxt = out.reshape(src.shape[0], *([nb] * (level + 1)), *out.shape[2:])
        while len(self._levels) <= level + 1:
            self._levels.append(None)
        self._levels[level + 1] = nxt
        for key in self._keys_of_level(level + 1):
            if key not in self._assigned:
                self.data[key] = self._node_view(nxt, key)

    # ---- dict protocol ----------------------------------------------------------------------------------------
    def __setitem__(self, key: str, value: torch.Tensor) -> None:
        self._assigned.add(key)
        self.data[key] = value

    def __getitem__(self, key: str) -> torch.Tensor:
        """Lazy node access with the reference's error behaviour (src/ptwt/packets.py:318-359, :622-668)."""
        if self.maxlevel is None:
            raise ValueError("The wavelet packet tree must be initialized via 'transform' before its values can be accessed!")
        if key not in self.data:
            if len(key) > self.maxlevel:
                raise KeyError(
                    f"The requested level {len(key)} with key '{key}' is too large and cannot be accessed! This "
                    f"wavelet packet tree is initialized with maximum level {self.maxlevel}."
                )
            if key == "":
                raise ValueError(
                    "The requested root of the packet tree cannot be accessed! The wavelet packet tree is not "
                    "properly initialized. Run `transform` before accessing tree values."
                )
            if any(c not in self._filter_keys for c in key):
                raise ValueError(f"Invalid key '{key}'. All chars in the key must be of the set {self._filter_keys}.")
            if self._layout is None or not self._levels:
                raise ValueError(
                    "The requested root of the packet tree cannot be accessed! The wavelet packet tree is not "
                    "properly initialized. Run `transform` before accessing tree values."
                )
            start = len(key) - 1
            while start > 0 and not all(k in self.data for k in self._keys_of_level(start)):
                start -= 1
            for level in range(start, len(key)):
                self._expand_level(level)
        return self.data[key]

    # ---- synthesis ----------------------------------------------------------------------------------------------
    def reconstruct(self):
        """Rebuild every inner node, and finally the input, from the leaves of ``maxlevel`` — one synthesis launch
        per level (src/ptwt/packets.py:190-238, :480-526).  Raises ``KeyError`` when a leaf is missing."""
        if self.maxlevel is None:
            root = self[""]  # raises the reference's ValueError for an uninitialised tree
            self.maxlevel = dwt_max_level(min(self._layout.fold(root).shape[1:]), filter_length(self.wavelet))
        _, _, rec_lo, rec_hi = host_taps(self.wavelet)
        tap_t = _fwt._tap_tensors(self.wavelet)
        flen = len(rec_lo)
        nb = len(self._bands)
        nd = self._ndim
        for level in reversed(range(self.maxlevel)):
            for node in self._keys_of_level(level):
                for child in self._bands:
                    if node + child not in self.data:
                        raise KeyError(f"Key {node + child} not found")
            children = self._level_buffer(level + 1)  # [B, nb^(level+1), M..]
            flat = children.reshape(-1, nb, *children.shape[2 + level:])
            coef = flat.shape[2:]
            out_ext = [2 * m - flen + 2 for m in coef]
            if level > 0:
                # crop to the extents this level had in the analysis (one sample shorter for odd lengths)
                target = self._layout.fold(self[self._keys_of_level(level)[0]]).shape[1:]
                for a in range(nd):
                    if out_ext[a] != target[a]:
                        assert out_ext[a] == target[a] + 1, "padding error, please open an issue on github"
                        out_ext[a] = target[a]
            approx, details = flat[:, 0], [flat[:, s] for s in range(1, nb)]
            if torch.is_grad_enabled() and (flat.requires_grad or tap_t is not None):
                rec = _fwt._SynthesisLevel.apply(rec_lo, rec_hi, tuple(out_ext), *((tap_t[2], tap_t[3]) if tap_t else (None, None)), approx, *details)
            else:
                rec = _engine.ENGINE.synthesis(approx, details, rec_lo, rec_hi, out_ext)
            buf = rec.reshape(children.shape[0], *([nb] * level), *rec.shape[1:])
            while len(self._levels) <= level:
                self._levels.append(None)
            self._levels[level] = buf
            for key in self._keys_of_level(level):
                self._assigned.discard(key)
                self.data[key] = self._node_view(buf, key)
        return self


class WaveletPacket(_PacketTree):
    """One-dimensional wavelet packet tree (drop-in for ``ptwt.WaveletPacket``, src/ptwt/packets.py:67-360)."""

    _ndim = 1
    _bands = {"a": 0, "d": 1}

    def __init__(self, data: Optional[torch.Tensor], wavelet, *, mode="reflect", maxlevel: Optional[int] = None,
                 axis: Union[int, Sequence[int], None] = None, orthogonalization: str = "qr") -> None:
        if orthogonalization not in ("qr", "gramschmidt"):
            raise NotImplementedError
        self.orthogonalization = orthogonalization
        super().__init__(data, wavelet, mode, maxlevel, axis)
        self.axis = self._axes[0]

    @staticmethod
    def get_level(level: int, order: str = "freq") -> List[str]:
        """Paths of all nodes of ``level`` in frequency (Gray code) or natural order (src/ptwt/packets.py:273-310)."""
        if order == "freq":
            return _graycode(level, "a", "d")
        if order == "natural":
            return ["".join(p) for p in product("ad", repeat=level)]
        raise ValueError(f"Unsupported order '{order}'. Choose from 'freq' and 'natural'.")


class WaveletPacket2D(_PacketTree):
    """Two-dimensional wavelet packet tree (drop-in for ``ptwt.WaveletPacket2D``, src/ptwt/packets.py:362-771).

    Key chars: ``a`` approximation, ``h`` / ``v`` / ``d`` the (H, V, D) details of ``wavedec2`` — i.e. engine bands
    ``da`` / ``ad`` / ``dd``.  With ``separable=True`` the reference routes through ``fswavedec2`` and maps its
    ``"ad"`` band to ``h`` and ``"da"`` to ``v`` (src/ptwt/packets.py:592-620); the same mapping is kept here."""

    _ndim = 2
    _bands = {"a": 0, "h": 2, "v": 1, "d": 3}

    def __init__(self, data: Optional[torch.Tensor], wavelet, *, mode="reflect", maxlevel: Optional[int] = None,
                 axes: Union[Sequence[int], None] = None, orthogonalization: str = "qr", separable: bool = False) -> None:
        if orthogonalization not in ("qr", "gramschmidt"):
            raise NotImplementedError
        self.orthogonalization = orthogonalization
        self.separable = separable
        if separable:
            self._bands = {"a": 0, "h": 1, "v": 2, "d": 3}
        super().__init__(data, wavelet, mode, maxlevel, axes)
        self.axes = self._axes

    @staticmethod
    def get_natural_order(level: int) -> List[str]:
        return ["".join(p) for p in product("ahvd", repeat=level)]

    @staticmethod
    def get_freq_order(level: int) -> List[List[str]]:
        """2-D frequency ordering: rows / columns follow the Gray-code order of the per-axis low/high paths
        (src/ptwt/packets.py:719-771; key char -> (row filter, column filter): a = ll, h = hl, v = lh, d = hh)."""
        split = {"a": ("l", "l"), "h": ("h", "l"), "v": ("l", "h"), "d": ("h", "h")}
        grid: Dict[str, Dict[str, str]] = {}
        for node in product("ahvd", repeat=level):
            row = "".join(split[c][0] for c in node)
            col = "".join(split[c][1] for c in node)
            grid.setdefault(row, {})[col] = "".join(node)
        order = _graycode(level, "l", "h") if level > 0 else [""]
        return [[grid[r][c] for c in order if c in grid[r]] for r in order if r in grid]

    @staticmethod
    def get_level(level: int, order: str = "freq"):
        if order == "freq":
            return WaveletPacket2D.get_freq_order(level)
        if order == "natural":
            return WaveletPacket2D.get_natural_order(level)
        raise ValueError(f"Unsupported order '{order}'. Choose from 'freq' and 'natural'.")
