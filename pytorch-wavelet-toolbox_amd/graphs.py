"""HIP-graph capture of whole transform calls (an MI355X-side extension; the reference has no counterpart).

A multi-level call is a handful of kernel launches plus the Python around them: level loop, plan lookup, ``torch.empty``, one ctypes
call per launch — 20-45 us of host time per call, which is what a call on a small batch costs in full (the kernels are done sooner).
Every launch of this package goes to PyTorch's CURRENT stream and nothing in a call synchronises or touches the host (string / host
wavelets), so a whole call — or any function made of such calls — can be recorded once into a HIP graph and replayed:

    fwd = ptwt_amd.capture(lambda t: ptwt_amd.wavedec2(t, "db4", level=3), x)      # records the launches for x's geometry
    coeffs = fwd(x_new)                                                              # one graph launch; same containers

Measured (MI355X, `tools/graph_probe.py`, ``waverec2(wavedec2(x))`` per iteration): 16 x 64^2 db2 level 3 47.3 -> 22.6 us, 8 x 256^2 db4
level 4 86.0 -> 39.6 us (the replay runs at the kernels' own time); calls that are GPU-bound anyway gain nothing (the reference's 2-D
speed-test shape, 32 x 1000^2 db5 level 5 periodic: 156 against 160 us).  Bit-identical to the eager call (the same kernels on the same data).
"""
from __future__ import annotations

from typing import Any, Callable

import torch

__all__ = ["capture", "CapturedCall"]


def _map_tensors(obj: Any, fn: Callable[[torch.Tensor], Any]) -> Any:
    if isinstance(obj, torch.Tensor):
        return fn(obj)
    if isinstance(obj, dict):
        return {k: _map_tensors(v, fn) for k, v in obj.items()}
    if isinstance(obj, tuple) and hasattr(obj, "_fields"):  # named tuples (WaveletDetailTuple2d)
        return type(obj)(*[_map_tensors(v, fn) for v in obj])
    if isinstance(obj, (list, tuple)):
        return type(obj)(_map_tensors(v, fn) for v in obj)
    return obj


class CapturedCall:
    """``fn(x)`` recorded into a HIP graph for the geometry (shape, strides, dtype, device) of ``example``.

    Calling the object copies its argument into the graph's static input buffer, replays the graph and returns the STATIC outputs —
    the same container structure ``fn`` returned, whose tensors live in the graph's memory pool and are overwritten by the next replay
    (``clone`` what must outlive it).  ``fn`` must not synchronise with the host: wavelets given by name or as host numbers are fine, a
    filter bank of GPU tensors is read back on every call and cannot be captured; gradients are not recorded (inference only).
    """

    def __init__(self, fn: Callable[[torch.Tensor], Any], example: torch.Tensor, warmup: int = 3):
        if not example.is_cuda:
            raise RuntimeError("ptwt_amd.capture: the example input must live on a ROCm device")
        self._fn = fn
        self.static_input = example.detach().clone()
        dev = example.device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.no_grad(), torch.cuda.stream(side):
            for _ in range(max(1, warmup)):  # plans, routing memos and the allocator's pools settle outside the capture
                fn(self.static_input)
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.static_output = fn(self.static_input)

    def __call__(self, x: torch.Tensor) -> Any:
        si = self.static_input
        if x.shape != si.shape or x.dtype != si.dtype or x.device != si.device:
            raise ValueError(f"captured for a {tuple(si.shape)} {si.dtype} tensor on {si.device}, got {tuple(x.shape)} {x.dtype} on {x.device}")
        if x.data_ptr() != si.data_ptr():
            si.copy_(x)
        self.graph.replay()
        return self.static_output

    def replay(self) -> Any:
        """Replay on whatever ``static_input`` holds (fill it in place to skip the copy of ``__call__``)."""
        self.graph.replay()
        return self.static_output

    def cloned(self, x: torch.Tensor) -> Any:
        """``self(x)`` with every output tensor cloned out of the graph's memory pool."""
        return _map_tensors(self(x), lambda t: t.clone())


def capture(fn: Callable[[torch.Tensor], Any], example: torch.Tensor, warmup: int = 3) -> CapturedCall:
    """Record ``fn(example)`` — any function of one device tensor made of this package's transforms — into a HIP graph; see
    :class:`CapturedCall`."""
    return CapturedCall(fn, example, warmup)
