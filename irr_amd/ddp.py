"""Synchronous data parallelism for the IRR-PWC step: one process per GPU, image pairs sharded by rank,
gradient all-reduce (mean) with RCCL over xGMI (``torch.distributed`` backend "nccl"; "gloo" on CPU for tests).

The reference has no multi-GPU path at all (DataParallel is commented out, main.py:47-53); this is the
MI355X-side addition required by BASELINE.json.  Design:

* all 124 gradients live in ONE flat fp32 arena (25.45 MB); ``param.grad`` are views into it, so a bucket
  is a contiguous slice and needs no flatten/unflatten copies;
* buckets are ordered by when their gradients become final during backward.  IRR-PWC shares its decoder
  weights over all pyramid levels, so almost every gradient is final only when the coarsest level has been
  back-propagated; the exceptions are the occlusion upsampler and ``conv_1x1_1`` (levels 5-6 only), which
  finish first and are reduced on a side stream while the rest of backward (incl. the correlation-backward
  kernels of levels 4..0) is still running;
* the two detached loss scalars are all-reduced inside the loss (``reduce_losses``) so the flow/occ balancing
  weights (losses.py:560-567) equal those of a single process on the global batch.
"""
from __future__ import annotations

from typing import Callable, Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

EARLY_PREFIXES = ("occ_shuffle_upsample.", "conv_1x1_1.")


def shard_batch(batch: dict, rank: int, world: int) -> dict:
    """rank r gets samples [r*B/world, (r+1)*B/world) of every tensor in the batch dict."""
    out = {}
    for k, v in batch.items():
        if torch.is_tensor(v) and v.dim() > 0:
            n = v.shape[0]
            assert n % world == 0, f"global batch {n} not divisible by world size {world}"
            per = n // world
            out[k] = v[rank * per:(rank + 1) * per]
        else:
            out[k] = v
    return out


def reduce_losses(group=None) -> Callable:
    """reduce_fn for MultiScaleEPE_PWC_Bi_Occ_upsample: SUM of (flow_loss, occ_loss) over ranks."""
    def fn(f_loss, o_loss):
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return f_loss, o_loss
        t = torch.stack([f_loss, o_loss])
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return t[0], t[1]
    return fn


class GradArena:
    """Flat gradient storage + bucketed, overlapped all-reduce."""

    def __init__(self, named_params: Iterable[Tuple[str, torch.nn.Parameter]], group=None,
                 early_prefixes: Sequence[str] = EARLY_PREFIXES, overlap: bool = True):
        named = [(n, p) for n, p in named_params if p.requires_grad]
        early = [(n, p) for n, p in named if n.startswith(tuple(early_prefixes))]
        late = [(n, p) for n, p in named if not n.startswith(tuple(early_prefixes))]
        self.group = group
        self.order = early + late
        dev = self.order[0][1].device
        total = sum(p.numel() for _, p in self.order)
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        self.buckets: List[Tuple[int, int]] = []
        self._bucket_of = {}
        self._pending: List[int] = []
        off = 0
        for bi, grp in enumerate((early, late)):
            start = off
            for _, p in grp:
                n = p.numel()
                p.grad = self.flat[off:off + n].view_as(p)
                self._bucket_of[id(p)] = bi
                off += n
            self.buckets.append((start, off))
        self._counts = [len(early), len(late)]
        self._works = []
        self.overlap = overlap and dev.type == "cuda"
        self._side = torch.cuda.Stream(device=dev) if self.overlap else None
        self._hooks = []
        self._reset()
        if self.world > 1:
            for _, p in self.order:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    @property
    def world(self) -> int:
        return dist.get_world_size(self.group) if (dist.is_available() and dist.is_initialized()) else 1

    def _reset(self):
        self._pending = list(self._counts)
        self._launched = [False, False]

    def zero_grad(self):
        """replaces optimizer.zero_grad(): one memset, gradients stay views of the arena"""
        self.flat.zero_()
        for _, p in self.order:
            if p.grad is None or p.grad.data_ptr() < self.flat.data_ptr() or \
                    p.grad.data_ptr() >= self.flat.data_ptr() + self.flat.numel() * 4:
                self._reattach()
                break
        self._reset()

    def _reattach(self):
        off = 0
        for _, p in self.order:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n

    def _launch(self, bi: int):
        if self._launched[bi] or self.world == 1:
            return
        self._launched[bi] = True
        s, e = self.buckets[bi]
        if e == s:
            return
        chunk = self.flat[s:e]
        if self.overlap:
            self._side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._side):
                self._works.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self._works.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _on_grad(self, p):
        bi = self._bucket_of[id(p)]
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            self._launch(bi)

    def enable_async_wgrad(self):
        """Route every weight/bias gradient of the MFMA conv nodes straight into this arena on a second HIP stream
        (irr_amd.conv.WgradSide): the wgrad launches then overlap the data-gradient chain instead of sitting on
        its critical path.  Autograd no longer sees those gradients, so buckets are reduced at sync()."""
        from . import conv
        self._side_lane = conv.WgradSide({id(p): p.grad for _, p in self.order})
        conv.SIDE = self._side_lane

    def disable_async_wgrad(self):
        from . import conv
        if getattr(self, "_side_lane", None) is not None and conv.SIDE is self._side_lane:
            conv.SIDE = None
        self._side_lane = None

    def sync(self):
        """call between backward() and optimizer.step(): flush, wait, average."""
        if getattr(self, "_side_lane", None) is not None:
            self._side_lane.join()
        if self.world == 1:
            return
        for bi in range(len(self.buckets)):
            self._launch(bi)
        for w in self._works:
            w.wait()
        self._works.clear()
        if self.overlap:
            torch.cuda.current_stream().wait_stream(self._side)
        self.flat.mul_(1.0 / self.world)


def broadcast_params(module: torch.nn.Module, src: int = 0, group=None) -> None:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    for p in module.parameters():
        dist.broadcast(p.data, src=src, group=group)
