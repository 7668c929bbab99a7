"""Synchronous data parallelism for the IRR-PWC step: one process per GPU, image pairs sharded by rank,
gradient all-reduce (mean) with RCCL over xGMI (``torch.distributed`` backend "nccl"; "gloo" on CPU for tests).

The reference has no multi-GPU path at all (DataParallel is commented out, main.py:47-53); this is the
MI355X-side addition required by BASELINE.json.  Design:

* all 124 gradients live in ONE flat fp32 arena (25.45 MB); ``param.grad`` are views into it, so a bucket
  is a contiguous slice and needs no flatten/unflatten copies;
* three buckets, ordered by when their gradients become final during backward: the occlusion upsampler and
  ``conv_1x1_1`` (levels 5-6 only) finish first and are reduced on a communication stream while the rest of backward
  (incl. the correlation-backward kernels of levels 4..0) is still running; the shared decoders / context / refinement
  networks are final after the coarsest level and are reduced under the backward of the feature pyramid; the pyramid
  itself is reduced at ``sync()``;
* the two detached loss scalars are all-reduced inside the loss (``reduce_losses``) so the flow/occ balancing
  weights (losses.py:560-567) equal those of a single process on the global batch.
"""
from __future__ import annotations

import os
import time
import weakref
from typing import Callable, Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

# Buckets in the order in which their gradients become FINAL during backward (IRR-PWC shares its decoders over the levels):
#   0 "early":   the occlusion upsampler and conv_1x1_1 are only used at levels 5-6, which backward visits first;
#   1 "shared":  everything else (decoders, context networks, refinement, conv_1x1) is final after the coarsest level;
#   2 "late":    the feature pyramid is back-propagated last (its nodes are the oldest of the tape).
EARLY_PREFIXES = ("occ_shuffle_upsample.", "conv_1x1_1.")
LATE_PREFIXES = ("feature_pyramid_extractor.",)


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


def collectives_on(group=None) -> bool:
    """True when the data-parallel collectives run: an initialised process group of more than one rank -- or of ONE rank under
    IRR_DDP_SINGLE_RANK=1 (test switch: the RCCL transport, its communicator and the stream choreography around it on a one-GPU
    box -- RCCL refuses two ranks on one device; an all-reduce over one rank is the identity, everything around it is real)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or bool(os.environ.get("IRR_DDP_SINGLE_RANK"))


def reduce_losses(group=None) -> Callable:
    """reduce_fn for MultiScaleEPE_PWC_Bi_Occ_upsample: SUM of (flow_loss, occ_loss) over ranks."""
    def fn(f_loss, o_loss):
        if not collectives_on(group):
            return f_loss, o_loss
        t = torch.stack([f_loss, o_loss])
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return t[0], t[1]
    return fn


class GradArena:
    """Flat gradient storage + bucketed all-reduce that starts while backward is still running.

    A bucket's all-reduce is enqueued on a communication stream the moment its LAST gradient contribution of the step has
    been enqueued -- wherever that contribution is produced: by autograd (``post_accumulate_grad`` hooks, main stream) or by
    the asynchronous weight-gradient lane (irr_amd.conv.WgradSide, its own stream, bypassing autograd).  How many
    contributions each parameter receives per step is a property of the model graph (shared decoders: one per level and
    use); it is LEARNED from the first backward pass (which reduces everything at ``sync()``) and counted down from then
    on.  With IRR-PWC that puts the occlusion-upsampler bucket under the backward of levels 4..0 (correlation backward,
    decoders) and the 21 MB shared-decoder bucket under the backward of the feature pyramid; only the pyramid's own 4 MB
    are reduced after the last kernel of backward."""

    MAX_CALIBRATION_ATTEMPTS = 3

    def __init__(self, named_params: Iterable[Tuple[str, torch.nn.Parameter]], group=None,
                 early_prefixes: Sequence[str] = EARLY_PREFIXES, late_prefixes: Sequence[str] = LATE_PREFIXES,
                 overlap: bool = True):
        named = [(n, p) for n, p in named_params if p.requires_grad]
        early = [(n, p) for n, p in named if n.startswith(tuple(early_prefixes))]
        late = [(n, p) for n, p in named if n.startswith(tuple(late_prefixes)) and not n.startswith(tuple(early_prefixes))]
        taken = {id(p) for _, p in early + late}
        mid = [(n, p) for n, p in named if id(p) not in taken]
        self.group = group
        self.order = early + mid + late
        dev = self.order[0][1].device
        total = sum(p.numel() for _, p in self.order)
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        self.buckets: List[Tuple[int, int]] = []
        self._bucket_of = {}
        self._members: List[List[int]] = []
        off = 0
        for bi, grp in enumerate((early, mid, late)):
            start = off
            ids = []
            for _, p in grp:
                n = p.numel()
                p.grad = self.flat[off:off + n].view_as(p)
                self._bucket_of[id(p)] = bi
                ids.append(id(p))
                off += n
            self.buckets.append((start, off))
            self._members.append(ids)
        me = weakref.ref(self)
        for _, p in self.order:                         # (irr_amd.harness: never build a second arena over these parameters)
            p.__dict__["_irr_arena"] = me
        self._works = []
        self.overlap = overlap and dev.type == "cuda" and not os.environ.get("IRR_DDP_NO_OVERLAP")      # (environment: diagnosis switch)
        self._side = torch.cuda.Stream(device=dev) if self.overlap else None
        self._side_lane = None
        self._hooks = []
        self._expected: Optional[dict] = None          # id(param) -> contributions per step (None: not calibrated yet)
        self._expected_queued: Optional[List[int]] = None   # per bucket: weight-gradient launches QUEUED on the lane per step
        self.calibration_mismatch = False
        self._calib_attempts = 0
        self.broken: Optional[str] = None              # set by a late contribution: every later sync() raises until recalibrate()
        self._seen = {}
        self._t0 = time.perf_counter()
        self.launch_log: List[Tuple[int, str]] = []    # (bucket, "backward" | "sync") of the last step -- tests / diagnostics
        self.launch_times: List[float] = []            # ms since zero_grad() at which each entry of launch_log was enqueued
        self._reset()
        if self.active:
            for _, p in self.order:
                self._hooks.append(p.register_post_accumulate_grad_hook(lambda p_: self._contribution(id(p_))))

    @property
    def world(self) -> int:
        return dist.get_world_size(self.group) if (dist.is_available() and dist.is_initialized()) else 1

    @property
    def active(self) -> bool:
        """the gradient all-reduce runs (collectives_on: more than one rank, or the single-rank test switch)"""
        return collectives_on(self.group)

    def _reset(self):
        self._seen = {pid: 0 for pid in self._bucket_of}
        self._launched = [False] * len(self.buckets)
        self._in_sync = False
        self._late = False
        self._queued = [0] * len(self.buckets)
        self._streams = [[] for _ in self.buckets]      # per bucket: the streams its contributions were enqueued on (_launch)
        self._t0 = time.perf_counter()
        self.launch_times = []
        if self._expected is not None:
            self._remaining = [sum(1 for pid in ids if self._expected.get(pid, 0) > 0) for ids in self._members]
        else:
            self._remaining = [-1] * len(self.buckets)
        self.launch_log = []
        from . import functional as _fn                 # (the cost-volume gradient launch times of the step that starts now)
        _fn.CORR_BWD_LAUNCH_TIMES.clear()

    def launch_report(self) -> dict:
        """of the step that has just run (call after sync(), before the next zero_grad()): when each bucket's all-reduce was enqueued
        and when the cost-volume gradient kernels were issued, ms since the step's zero_grad() on the host clock -- bucket 0 is
        enqueued before the FIRST of them (levels 4..0 are visited in that order): the overlap north_star asks for."""
        from . import functional as _fn
        return {"bucket_launches": [(b_, w_, t_) for (b_, w_), t_ in zip(self.launch_log, self.launch_times)],
                "corr_backward_launches_ms": [round((t - self._t0) * 1e3, 3) for t in _fn.CORR_BWD_LAUNCH_TIMES]}

    def recalibrate(self):
        """forget the learned contribution counts (call when the model graph changes: other loss heads, frozen layers) -- on
        EVERY rank before the same step: the next step is a calibration step (all buckets reduced at sync(), two extra small
        all-reduces), and the ranks must run it together"""
        self._expected = None
        self._expected_queued = None
        self.calibration_mismatch = False
        self._calib_attempts = 0
        self.broken = None
        self._reset()

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

    def _inside(self, t: torch.Tensor) -> bool:
        return self.flat.data_ptr() <= t.data_ptr() < self.flat.data_ptr() + self.flat.numel() * 4

    def adopt_grads(self):
        """Make every ``param.grad`` a view of the arena again WITHOUT assuming who cleared it: a stock optimizer's
        ``zero_grad()`` sets the gradients to None (torch's default, ``set_to_none=True``; the reference calls it once per step,
        runtime.py:164) -- such a parameter gets its zeroed slice back; a gradient that still is the arena's view is kept as it is
        (gradient accumulation over several backward passes); a foreign gradient tensor is copied into its slice.  Called by the
        drop-in route (irr_amd.harness) at the start of every training forward pass."""
        grads = [p.grad for _, p in self.order]
        if all(g is None for g in grads):
            self.flat.zero_()
            self._reattach()
            return
        off = 0
        with torch.no_grad():
            for (_, p), g in zip(self.order, grads):
                n = p.numel()
                if g is None:
                    self.flat[off:off + n].zero_()
                    p.grad = self.flat[off:off + n].view_as(p)
                elif not (self._inside(g) and g.data_ptr() == self.flat.data_ptr() + 4 * off):
                    self.flat[off:off + n].copy_(g.reshape(-1))
                    p.grad = self.flat[off:off + n].view_as(p)
                off += n

    def readopt_routed(self, params) -> None:
        """End of a backward pass on the drop-in route (WgradSide.on_join): ``params`` received contributions that the lane
        accumulated straight into their arena slices.  If the caller cleared the gradients between forward and backward
        (``optimizer.zero_grad()`` with set_to_none after the forward pass), ``.grad`` is None -- or a fresh tensor autograd created
        for a contribution that did not go through the lane -- while the routed sum sits in the slice: give the parameter its slice
        back (adding a foreign tensor into it), so ``optimizer.step()`` sees the complete gradient."""
        base = self.flat.data_ptr()
        offs = getattr(self, "_offsets", None)
        if offs is None:
            offs, off = {}, 0
            for _, p in self.order:
                offs[id(p)] = off
                off += p.numel()
            self._offsets = offs
        with torch.no_grad():
            for p in params:
                off = offs.get(id(p))
                if off is None:
                    continue
                g = p.grad
                if g is not None and g.data_ptr() == base + 4 * off:
                    continue
                view = self.flat[off:off + p.numel()].view_as(p)
                if g is not None:
                    view.add_(g)
                p.grad = view

    def _launch(self, bi: int):
        if self._launched[bi] or not self.active:
            return
        self._launched[bi] = True
        self.launch_log.append((bi, "sync" if self._in_sync else "backward"))
        self.launch_times.append(round((time.perf_counter() - self._t0) * 1e3, 3))
        s, e = self.buckets[bi]
        if e == s:
            return
        chunk = self.flat[s:e]
        # Every stream that enqueued a contribution to this bucket since zero_grad() (_contribution: the autograd hook's stream -- the
        # occlusion branch of the coarse levels runs its backward nodes on a second stream, irr_pwc.py -- or the stream an inline
        # lane folded on) plus the stream this call runs on.  Waiting for a stream waits for everything enqueued on it so far: a
        # superset of the bucket's contributions, never less (with the last contribution on the main stream the earlier
        # ones of the branch stream were not waited for).
        cur = torch.cuda.current_stream() if chunk.is_cuda else None
        contributors = [s_ for s_ in self._streams[bi] if s_ != cur] if cur is not None else []
        if cur is not None and not torch.cuda.is_current_stream_capturing():
            main = torch.cuda.default_stream(chunk.device)
            if main != cur and main not in contributors:
                contributors.append(main)
        if self.overlap:
            # The lane's tail at this moment IS the bucket's last routed contribution: _on_queue() flushed the lane the moment the
            # bucket's last weight-gradient launch of the step was queued, so the fold that completes the bucket is the last
            # thing on the lane -- the all-reduce does not wait for later levels' launches that the lagging lane still holds.
            self._side.wait_stream(cur)
            for s_ in contributors:
                self._side.wait_stream(s_)
            if self._side_lane is not None and self._side_lane.stream is not None:
                self._side.wait_stream(self._side_lane.stream)
            with torch.cuda.stream(self._side):
                self._works.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            # no communication stream: the collective is ordered after the CURRENT stream only (the backend's own rule), so the
            # current stream waits for the other contributors first
            for s_ in contributors:
                cur.wait_stream(s_)
            if cur is not None and self._side_lane is not None and self._side_lane.stream is not None:
                cur.wait_stream(self._side_lane.stream)
            self._works.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _contribution(self, pid: int):
        """one gradient contribution to parameter ``pid`` has just been enqueued (autograd hook or lane launch)"""
        bi = self._bucket_of.get(pid)
        if bi is None:
            return
        self._seen[pid] += 1
        if self.flat.is_cuda:
            st = torch.cuda.current_stream()
            if st not in self._streams[bi]:
                self._streams[bi].append(st)
        if self._expected is None:
            return
        if self._launched[bi]:
            # Not raised here: this runs inside an autograd hook / lane callback on THIS rank only, and the other ranks have
            # already enqueued the matching all-reduce -- they would hang in sync().  The step's collectives are completed
            # as usual and sync() raises once they are matched (the bucket's sum lacks this contribution).
            self._late = True
            return
        if self._seen[pid] == self._expected.get(pid, 0):
            self._remaining[bi] -= 1
            if self._remaining[bi] == 0:
                self._launch(bi)

    def _on_lane(self, weight, bias):
        self._contribution(id(weight))
        if bias is not None:
            self._contribution(id(bias))

    def _on_queue(self, weight, bias):
        """a weight-gradient launch for ``weight`` has just been QUEUED on the lane (its partial images are folded, and the
        contribution reported through _on_lane, only when the lane's fold batch is flushed -- up to 40 launches later).  When it
        is the bucket's last launch of the step, flush now: the bucket's all-reduce then waits for exactly that fold."""
        bi = self._bucket_of.get(id(weight))
        if bi is None:
            return
        self._queued[bi] += 1
        if (self._expected_queued is not None and not self._launched[bi]
                and self._queued[bi] == self._expected_queued[bi] and self._side_lane is not None):
            self._side_lane.flush()

    def enable_async_wgrad(self):
        """Route every weight/bias gradient of the MFMA conv nodes straight into this arena on a second HIP stream
        (irr_amd.conv.WgradSide): the wgrad launches then overlap the data-gradient chain instead of sitting on
        its critical path.  Autograd no longer sees those gradients; the lane reports each contribution instead."""
        from . import conv
        self._side_lane = conv.WgradSide([(p, p.grad) for _, p in self.order])
        if self.active:
            self._side_lane.on_launch = self._on_lane
            self._side_lane.on_queue = self._on_queue
        conv.SIDE = self._side_lane
        self.recalibrate()

    def enable_direct_wgrad(self):
        """The same routing WITHOUT a second stream: the weight-gradient launches stay on the current stream but accumulate
        straight into this arena (no per-use gradient tensors, no autograd accumulation adds, batched folds)."""
        from . import conv
        self._side_lane = conv.WgradSide([(p, p.grad) for _, p in self.order], inline=True)
        if self.active:
            self._side_lane.on_launch = self._on_lane
            self._side_lane.on_queue = self._on_queue
        conv.SIDE = self._side_lane
        self.recalibrate()

    def disable_async_wgrad(self):
        from . import conv
        if self._side_lane is not None:
            self._side_lane.join()                      # (folds whatever the lane still holds as partial images)
        if self._side_lane is not None and conv.SIDE is self._side_lane:
            conv.SIDE = None
        self._side_lane = None
        self.recalibrate()

    def sync(self):
        """call between backward() and optimizer.step(): flush, wait, average."""
        if self._side_lane is not None:
            self._side_lane.join()
        if not self.active:
            return
        self._in_sync = True
        for bi in range(len(self.buckets)):
            self._launch(bi)
        for w in self._works:
            w.wait()
        self._works.clear()
        if self.overlap:
            torch.cuda.current_stream().wait_stream(self._side)
        self.flat.mul_(1.0 / self.world)
        if self._late:
            # The calibrated counts stay as they are (this rank keeps starting its buckets where the others do, so the collective
            # sequences of all ranks still match) and the arena refuses to go on: the condition is rank-local knowledge, a rank
            # that silently switched to reduce-at-sync mode would deadlock the others.
            self.broken = ("a gradient contribution arrived after its bucket's all-reduce was started: the model graph differs "
                           "from the calibrated one (the reduced gradients of that step were incomplete) -- call "
                           "GradArena.recalibrate() on EVERY rank before the same step")
        if self.broken is not None:
            raise RuntimeError(self.broken)
        if self._expected is None and self._calib_attempts < self.MAX_CALIBRATION_ATTEMPTS:
            # calibration step: every bucket was reduced here.  The learned counts decide WHEN each rank starts a bucket's
            # collective, so they must agree on every rank (ranks whose graphs differ would start the buckets in different
            # orders and deadlock RCCL): compare them; on a mismatch this step stays the reduce-at-sync step it was and the next
            # one is tried again (a first step may differ for transient reasons), MAX_CALIBRATION_ATTEMPTS times in all -- then the
            # verdict is cached: reduce-at-sync mode without further comparison all-reduces / host syncs until recalibrate().
            self._calib_attempts += 1
            counts = torch.tensor([self._seen[id(p)] for _, p in self.order] + list(self._queued), dtype=torch.int64,
                                  device=self.flat.device)
            lo, hi = counts.clone(), counts.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
            if bool((lo == hi).all()):
                self._expected = dict(self._seen)
                self._expected_queued = list(self._queued)
            else:
                self.calibration_mismatch = True
                import warnings
                warnings.warn("GradArena: gradient-contribution counts differ between ranks -- buckets stay reduced at sync()"
                              + (" until recalibrate()" if self._calib_attempts >= self.MAX_CALIBRATION_ATTEMPTS else ""))


def broadcast_params(module: torch.nn.Module, src: int = 0, group=None) -> None:
    if not collectives_on(group):
        return
    for p in module.parameters():
        dist.broadcast(p.data, src=src, group=group)
