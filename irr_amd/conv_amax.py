"""Amax slots of the fp16x2 ("h2") conv kernels (include/irr_hip.h, section "h2"; csrc/x3_split.h).

fp16 has a 5-bit exponent, so the h2 kernels scale every operand tensor by a power of two derived -- on the device, no host
round trip -- from max |.| of the whole tensor.  The maxima live in small float32 device tensors ("slots"); an operand assembled
from several producers (a DenseNet buffer) carries one slot per part and the consumer takes the maximum over a run of slots.
A slot is filled either by the producing launch itself (``y_amax`` of irr_conv2d_fwd_h2: an atomic max in its epilogue) or by one
pass over the tensor (``measure``: irr_amax_f32, HBM-bound)."""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import hip
from .conv_pack import LAUNCHES

POOL_SLOTS = int(os.environ.get("IRR_AMAX_POOL", "16384"))      # (environment: diagnosis switch, profiles/NOTES.md C.5)
_POOLS = {}
_STREAMS = {}        # device index -> streams whose kernels use slots (Amax.zeros callers, the weight-gradient lane: note_stream)


def note_stream(stream) -> None:
    """``stream`` launches kernels that read or max into slots.  A pool tensor is allocated on ONE stream; the caching allocator
    orders its reuse after that stream only.  Every other stream is recorded on the pool (Tensor.record_stream), so that a retired
    pool -- freed on the host when its last Amax goes -- is not handed out again while kernels queued on another stream still
    touch its slots (a silently corrupted small tensor or a wrong power-of-two scale, once per ~4096 slots)."""
    key = stream.device.index if stream.device.index is not None else torch.cuda.current_device()
    known = _STREAMS.setdefault(key, [])
    if stream in known:
        return
    known.append(stream)
    pool = _POOLS.get(key)
    if pool is not None and pool[2] != stream:
        pool[0].record_stream(stream)


class Amax:
    """``n`` consecutive float32 slots of ``slots`` starting at ``first``: their maximum bounds |t| of some tensor t"""
    __slots__ = ("slots", "first", "n")

    def __init__(self, slots: torch.Tensor, first: int = 0, n: int = 1):
        assert slots.dtype == torch.float32 and slots.is_contiguous() and 0 <= first and first + n <= slots.numel() and n >= 1
        self.slots, self.first, self.n = slots, int(first), int(n)

    @staticmethod
    def zeros(device, n: int = 1) -> "Amax":
        """n fresh zeroed slots.  They are carved from a zero-filled pool tensor per device (one fill launch per POOL_SLOTS slots
        instead of one per autograd node, ~100 per step); a slot is handed out once and never reused, an exhausted pool is
        replaced (the old one lives as long as any Amax refers to it)."""
        if torch.cuda.is_current_stream_capturing():          # a captured step owns its slots: the replay must find them zeroed
            return Amax(torch.zeros(n, device=device, dtype=torch.float32), 0, n)
        key = device.index if device.index is not None else torch.cuda.current_device()
        pool = _POOLS.get(key)
        cur = torch.cuda.current_stream(device)
        note_stream(cur)
        if pool is None or pool[1] + n > pool[0].numel():
            t = torch.zeros(max(POOL_SLOTS, n), device=device, dtype=torch.float32)
            filled = torch.cuda.Event()
            filled.record(cur)                                  # the fill is ordered on THIS stream only
            for s_ in _STREAMS.get(key, ()):                    # (see note_stream)
                if s_ != cur:
                    t.record_stream(s_)
            pool = _POOLS[key] = [t, 0, cur, filled]
        elif pool[2] != cur:
            # slots handed to a node on ANOTHER stream (a branch stream, a warm-up side stream): nothing else orders their first
            # use after the pool's zero fill -- a slot could be maxed before the fill lands, or zeroed after
            cur.wait_event(pool[3])
        first = pool[1]
        pool[1] += n
        return Amax(pool[0], first, n)

    def ptr(self) -> int:
        return self.slots.data_ptr() + 4 * self.first

    def sub(self, i: int, n: int = 1) -> "Amax":
        return Amax(self.slots, self.first + i, n)


def measure(x: torch.Tensor, into: Optional[Amax] = None) -> Amax:
    """fold max |x| into a slot (a fresh zeroed one when ``into`` is None); x: (B, C, H, W) with dense planes"""
    B, C, H, W = x.shape
    a = into if into is not None else Amax.zeros(x.device, 1)
    LAUNCHES["amax"] += 1
    hip.call("irr_amax_f32", hip.ptr(x), B, C * H * W, hip.bs(x), a.ptr(), hip.stream())
    return a
