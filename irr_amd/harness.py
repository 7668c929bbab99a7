"""The drop-in route: what a model of this package arranges FOR ITSELF when it is trained by a loop that knows nothing about
this package -- the reference's ``TrainingEpoch._step`` (runtime.py:158-189):

    optimizer.zero_grad()                                  # torch.optim.Adam built by configuration.py:488-573
    loss_dict, output_dict = model_and_loss(example_dict)
    assert not isnan(training_loss.item())
    training_loss.backward()
    optimizer.step()

Nothing in that loop calls GradArena / FusedAdam / TrainStep, so on its own the model would run on the plain autograd route: one
freshly allocated gradient tensor per weight USE (the decoders are shared over five levels and two branches), ~1 000
accumulation / zero-fill launches per step, every weight-gradient kernel on the critical path of backward, ~250 single weight-pack
launches after each optimizer step.  ``auto_install(model)`` (called at the start of every training forward pass) gives that loop
the same machinery ``bench.py``'s own step uses, without any change on the caller's side:

* a flat gradient arena (``irr_amd.ddp.GradArena``) whose slices ARE the parameters' ``.grad`` -- re-adopted every forward pass,
  whatever ``optimizer.zero_grad()`` did to them (``GradArena.adopt_grads``);
* the asynchronous weight-gradient lane (``irr_amd.conv.WgradSide``) accumulating straight into those slices; the lane joins
  the caller's stream by itself at the end of every backward pass (final callback of the autograd engine), so
  ``optimizer.step()`` -- any ``torch.optim`` optimizer -- reads complete gradients;
  Gradients cleared BETWEEN forward and backward (``out = model(x); opt.zero_grad(); loss.backward()``) are re-attached at that
  join (``GradArena.readopt_routed``); a backward pass that raised is cleaned up at the next forward pass (``WgradSide.abandon``).
  Not supported on this route: ``torch.autograd.grad(loss, parameters)`` -- the routed gradients bypass autograd and come back as
  None; use ``IRR_AUTO_LANE=0`` (or ``harness.set_enabled(False)``) for that.
* weight-pack caches that notice in-place parameter updates through the tensors' version counters and refresh all packed
  copies with ONE launch (``irr_amd.conv._announce_rewrite``).

Not installed when the caller has set up a lane / arena itself, when ``torch.distributed`` runs with more than one rank (a
wrapper like DistributedDataParallel must see the gradients in autograd; use ``GradArena`` + ``TrainStep(grad_sync=arena.sync)``
there), or with ``IRR_AUTO_LANE=0`` (A/B switch: the plain autograd route).
"""
from __future__ import annotations

import os
import weakref

import torch

from . import conv as _conv

_ENABLED = os.environ.get("IRR_AUTO_LANE", "1") != "0"
_STATE = weakref.WeakKeyDictionary()          # model -> (signature, GradArena)


def enabled() -> bool:
    return _ENABLED


def set_enabled(on: bool) -> None:
    global _ENABLED
    _ENABLED = bool(on)


def installed(model) -> bool:
    st = _STATE.get(model)
    return st is not None and _conv.SIDE is st[1]._side_lane


def uninstall(model) -> None:
    st = _STATE.pop(model, None)
    if st is not None and st[1]._side_lane is not None:
        st[1].disable_async_wgrad()


def _drop_lane(lane) -> None:
    """the model died: its lane must not stay installed (it holds the model's gradient arena alive)"""
    if _conv.SIDE is lane:
        _conv.SIDE = None


def auto_install(model: torch.nn.Module) -> None:
    """see the module docstring; cheap when already installed (one pass over the parameter list)"""
    if not _ENABLED or not model.training or not torch.is_grad_enabled():
        return
    st = _STATE.get(model)
    side = _conv.SIDE
    mine = st[1]._side_lane if st is not None else None
    if side is not None and side is not mine:
        if not getattr(side, "auto", False):
            return                             # the caller installed its own lane (GradArena.enable_*_wgrad)
        side.join()                            # another model's automatic lane: it must not stay installed while THIS model runs
        _conv.SIDE = side = None
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    if not named or not all(p.is_cuda for _, p in named):
        return
    for _, p in named:                         # an arena the caller built over these parameters owns their gradients
        ref = p.__dict__.get("_irr_arena")
        other = ref() if ref is not None else None
        if other is not None and (st is None or other is not st[1]):
            if st is not None:
                _STATE.pop(model, None)
                if _conv.SIDE is mine:
                    _conv.SIDE = None
            return
    capturing = torch.cuda.is_current_stream_capturing()
    sig = tuple((id(p), p.device.index) for _, p in named)
    if st is None or st[0] != sig:
        if capturing:
            return
        if st is not None:
            uninstall(model)
        from .ddp import GradArena
        keep = {id(p): p.grad for _, p in named if p.grad is not None}      # GradArena() re-points .grad at zeroed slices
        arena = GradArena(named)
        with torch.no_grad():
            for _, p in named:
                if id(p) in keep:
                    p.grad.copy_(keep[id(p)])
        st = (sig, arena)
        _STATE[model] = st
    arena = st[1]
    if arena._side_lane is None:
        if capturing:
            return
        arena.enable_async_wgrad()             # (installs the lane as conv.SIDE)
        arena._side_lane.auto = True
        arena._side_lane.on_join = arena.readopt_routed       # gradients cleared between forward and backward (module docstring)
        weakref.finalize(model, _drop_lane, arena._side_lane)
    elif _conv.SIDE is not arena._side_lane:
        _conv.SIDE = arena._side_lane
    if not capturing:
        lane = arena._side_lane
        # (not while a backward pass is RUNNING on this thread -- a forward executed inside it, e.g. torch.utils.checkpoint's
        # recomputation or a hook that calls the model, finds the live pass's queued launches and fold jobs, which are not
        # leftovers: abandoning them would lose gradients silently)
        in_backward = torch._C._current_graph_task_id() != -1
        if lane is not None and lane.stale() and not in_backward:
            # the previous backward pass raised (OOM, an assertion in a hook): autograd skipped its final callbacks, so the lane
            # was never joined and still holds launches / fold jobs of the failed pass
            lane.abandon()
        if not in_backward:                    # (a live pass's routed sums sit in the slices: nothing to re-adopt, nothing to zero)
            arena.adopt_grads()
