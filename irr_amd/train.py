"""One optimisation step of IRR-PWC -- the build's counterpart of the reference's
``TrainingEpoch._step`` (runtime.py:131-194) + ``ModelAndLoss.forward`` (configuration.py:45-62) with
Adam(lr=1e-4, weight_decay=4e-4) (scripts/IRR-PWC_flyingChairsOcc.sh:29-31):

    [augmentation] -> zero_grad -> forward -> loss -> backward -> [grad all-reduce] -> NaN assert -> optimizer.step

(the reference asserts between loss and backward; TrainStep(check_nan="before_backward") keeps that placement, the default
asserts before the optimizer step from a pinned host copy of the loss -- same exception, weights never touched on NaN, no drain of
the GPU pipeline between forward and backward)
"""
from __future__ import annotations

import math
import os
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import conv as _conv
from . import conv_pack as _conv_pack


_GRAD_FINITE_LOG = bool(os.environ.get("IRR_GRAD_FINITE_LOG"))
_GRAD_STEP = [0]


def _grad_finite_report(module) -> None:
    """IRR_GRAD_FINITE_LOG=1 (diagnosis switch): after backward + gradient sync, list the parameters whose gradient holds a non-finite
    value (one device-side reduction per parameter, one host read per step) -- the first step that prints names the layer in which a
    NaN / inf entered the optimisation step."""
    import sys
    named = [(n, p) for n, p in module.named_parameters() if p.grad is not None]
    flags = torch.stack([torch.isfinite(p.grad).all() for _, p in named])
    amax = torch.stack([p.grad.abs().max() for _, p in named])
    _GRAD_STEP[0] += 1
    if not bool(flags.all()):
        bad = [(n, float(a)) for (n, _), f, a in zip(named, flags.tolist(), amax.tolist()) if not f]
        good = [n for (n, _), f in zip(named, flags.tolist()) if f]
        print(f"[grad log] step {_GRAD_STEP[0]}: {len(bad)} of {len(named)} gradients non-finite; FINITE: " + " ".join(n.replace("_model.", "") for n in good),
              file=sys.stderr, flush=True)
    elif os.environ.get("IRR_GRAD_FINITE_LOG") == "2":
        top = sorted(((float(a), n) for (n, _), a in zip(named, amax.tolist())), reverse=True)[:3]
        print(f"[grad log] step {_GRAD_STEP[0]}: all finite, largest |g| {top}", file=sys.stderr, flush=True)


class ModelAndLoss(nn.Module):
    """configuration.py:20-62: chains model -> loss, owns the ``_model.*`` state_dict namespace."""

    def __init__(self, args, model, training_loss, evaluation_loss=None):
        super().__init__()
        self._model = model
        self._training_loss = training_loss
        self._evaluation_loss = evaluation_loss if evaluation_loss is not None else training_loss

    @property
    def model(self):
        return self._model

    @property
    def training_loss(self):
        return self._training_loss

    def forward(self, example_dict):
        output_dict = self._model(example_dict)
        loss = self._training_loss if self.training else self._evaluation_loss
        return loss(output_dict, example_dict), output_dict


class TrainStep:
    """Holds model+loss+optimizer and runs reference-equivalent steps on device-resident batches."""

    def __init__(self, model_and_loss: ModelAndLoss, optimizer: torch.optim.Optimizer, training_key: str = "total_loss",
                 grad_sync=None, check_nan=True, augmentation=None, input_grads: bool = True):
        """augmentation: optional callable(example_dict) -> example_dict applied under ``no_grad`` to the device-resident batch
        before the forward pass, where the reference's ``_step`` runs its GPU augmentation (runtime.py:151-153), e.g.
        ``irr_amd.augment.RandomAffineFlowOcc`` (augmentations.py:368-653).
        check_nan: True / "before_step" -- the reference's per-step assertion (runtime.py:182-183) evaluated BEFORE THE OPTIMIZER
        STEP: the loss is copied to pinned host memory right after the forward pass, backward is enqueued, and the host reads the
        value (already there by then) before optimizer.step() -- same exception, no update of the weights on NaN, but no drain of
        the GPU pipeline between forward and backward.  "before_backward": the reference's exact placement (``.item()`` before
        ``backward()``; the GPU idles while the host re-issues the backward pass).  False: no check.
        input_grads: True = the reference's ``_step`` literally (runtime.py:158-162 marks every INPUT tensor ``requires_grad_(True)``, a
        pre-0.4 Variable idiom): backward also produces d loss / d image for both images -- through the warps of the raw images, the five
        image resizes of the refinement levels and the first pyramid convolution -- which nothing reads.  False: the inputs are plain
        tensors; losses, parameter gradients and the update are bit-identical (tests/test_train_gpu.py), ``input1.grad`` stays None."""
        self.model_and_loss = model_and_loss
        self.optimizer = optimizer
        self.training_key = training_key
        self.grad_sync = grad_sync              # callable() run between backward and optimizer.step (data parallel)
        self.augmentation = augmentation
        self.input_grads = bool(input_grads)
        if check_nan not in (True, False, "before_step", "before_backward"):
            raise ValueError(check_nan)
        self.check_nan = "before_step" if check_nan is True else check_nan
        self._loss_host = None                  # pinned scalar for the deferred check

    def __call__(self, example_dict: Dict[str, torch.Tensor]):
        if self.augmentation is not None:        # runtime.py:151-153
            with torch.no_grad():
                example_dict = self.augmentation(example_dict)
        for key, t in example_dict.items():      # runtime.py:158-162
            if "input" in key:
                t.requires_grad_(self.input_grads)
            elif "target" in key:
                t.requires_grad_(False)
        self.optimizer.zero_grad()
        loss_dict, output_dict = self.model_and_loss(example_dict)
        training_loss = loss_dict[self.training_key]
        copied = None
        if self.check_nan == "before_backward":   # runtime.py:182-183 as placed there (device->host sync between forward and backward)
            assert not math.isnan(training_loss.item()), "training_loss is NaN"
        elif self.check_nan == "before_step" and not torch.cuda.is_current_stream_capturing():
            if self._loss_host is None:
                self._loss_host = torch.empty((), dtype=torch.float32, pin_memory=True)
            self._loss_host.copy_(training_loss.detach(), non_blocking=True)
            copied = torch.cuda.Event()
            copied.record()
        training_loss.backward()
        if self.grad_sync is not None:
            self.grad_sync()
        elif _conv.SIDE is not None:
            # weight gradients routed past autograd (enable_async_wgrad / enable_direct_wgrad) are complete only after the lane's
            # deferred folds have run and the lane is joined: without a grad_sync (= GradArena.sync) that happens here
            _conv.SIDE.join()
        if _GRAD_FINITE_LOG:                      # diagnosis (profiles/NOTES.md C.5): which parameter's gradient is the first non-finite one
            _grad_finite_report(self.model_and_loss)
        if copied is not None:                    # the value left the device long ago: this wait does not stall the pipeline
            copied.synchronize()
            if math.isnan(float(self._loss_host)) and _conv._CHECK_FINITE in ("async", "slots"):
                _conv.dump_finite_log()
            assert not math.isnan(float(self._loss_host)), "training_loss is NaN"
        self.optimizer.step()
        _conv.WEIGHT_EPOCH[0] += 1                # every packed weight copy is stale now: they are refreshed by ONE launch
        return loss_dict, output_dict, example_dict["input1"].shape[0]


class GraphedTrainStep:
    """The same optimisation step captured ONCE into a hipGraph and replayed (MI355X-side addition; the reference has no
    counterpart).  One step of IRR-PWC is ~2 000 kernel launches, many of them on 6x7 ... 24x28 pyramid levels that finish
    faster than the host can issue them; a replay is a single launch of the whole dependency graph -- both HIP streams of
    the step (main + asynchronous weight-gradient lane) become branches of it.

    Requirements (checked or arranged here): static shapes; inputs are copied into static buffers; the optimizer keeps
    its step count on the device (``FusedAdam(capturable=True)``); nothing inside the step reads device data on the host --
    the reference's per-step NaN assertion (runtime.py:182-183) is evaluated on the captured loss right after the replay.
    Data-parallel runs (world > 1) use the eager ``TrainStep``: the gradient all-reduce is not captured.
    Any weight-gradient routing works (asynchronous lane, same-stream ``enable_direct_wgrad``, plain autograd).  Round 3 had found
    replays captured WITHOUT the lane drifting (~1e-2 over ten steps) or producing NaN and refused them; round 4 traced it to
    ``hipMemsetAsync`` inside the capture (the zero fill of the warp backward's scatter target): as a graph memset node it was not
    ordered against the kernel that had just read the recycled allocation -- the lane merely kept every tensor alive until the end
    of backward, so nothing was recycled.  The library now zero-fills with a kernel (csrc/common.h: irr_zero_async;
    profiles/r4_graph_bisect.txt), and tests/test_train_gpu.py replays all three routings against eager steps.

    The optimizer's hyper-parameters (lr, betas, eps, weight decay) are kernel ARGUMENTS of the captured launch: after changing
    one of them (the reference trains with MultiStepLR, scripts/IRR-PWC_flyingChairsOcc.sh) the next call re-captures
    (``FusedAdam.hyper()`` is compared with the captured values)."""

    def __init__(self, step: TrainStep, warmup: int = 2):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            raise ValueError("GraphedTrainStep is single-process only: the data-parallel gradient all-reduce is not captured "
                             "(use TrainStep with GradArena.sync)")
        if getattr(step.optimizer, "capturable", False) is not True:
            raise ValueError("GraphedTrainStep needs an optimizer whose step count lives on the device (FusedAdam(capturable=True))")
        if warmup < 2:
            # step 1 builds the packed weights one by one, step 2 builds the job table of the batched pack launch (a pageable
            # host-to-device copy): neither may happen inside the capture
            raise ValueError("GraphedTrainStep needs warmup >= 2")
        self.step = step
        self.warmup = warmup
        self.check_nan = bool(step.check_nan)
        self.graph = None
        self._pinned = None                       # what the captured repack launch touches (conv_pack.pin_all)
        self.hyper = None
        self.static_in: Dict[str, torch.Tensor] = {}
        self.result = None

    def _capture(self, example_dict):
        # (detached: a caller's tensor may already require grad -- the static copies must be leaves of their own)
        self.static_in = {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in example_dict.items()}
        eager = TrainStep(self.step.model_and_loss, self.step.optimizer, self.step.training_key, self.step.grad_sync,
                          check_nan=False, augmentation=self.step.augmentation)
        # The batched repack launch of the captured step covers EVERY conv weight registered on the device -- other models of the
        # process included (conv_pack._PackRegistry.pin): drop what is garbage already, then hold the rest for the life of the graph
        import gc
        gc.collect()
        self._pinned = _conv_pack.pin_all() if not os.environ.get("IRR_GRAPH_NO_PIN") else None      # (diagnosis switch: the fault of tests/test_train_gpu.py::test_graphed_step_survives_the_death_of_another_model)
        snap = self.step.optimizer.snapshot()         # the warm-up steps below must not count as training steps
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream(device=cur.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):                 # warm-up off the default stream: allocator pools, caches, autograd
            for _ in range(self.warmup):
                eager(self.static_in)
        cur.wait_stream(side)
        self.step.optimizer.restore(snap)
        torch.cuda.synchronize()
        if self._pinned is not None:
            self._pinned = _conv_pack.pin_all()       # (+ what the warm-up registered: the table the capture will reuse)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):            # (records the step; nothing executes until replay())
            self.result = eager(self.static_in)

    def __call__(self, example_dict: Dict[str, torch.Tensor]):
        if self.graph is not None and self.hyper != self.step.optimizer.hyper():
            self.graph = None                                   # lr schedule moved: the captured kernel arguments are stale
        if self.graph is None:
            self._capture(example_dict)
            self.hyper = self.step.optimizer.hyper()
        else:
            with torch.no_grad():
                for k, v in example_dict.items():
                    if torch.is_tensor(v) and v.data_ptr() != self.static_in[k].data_ptr():
                        self.static_in[k].copy_(v, non_blocking=True)
        self.graph.replay()
        # the captured FusedAdam kernel rewrote the weights behind autograd's back (no version counter moves during a replay):
        # every packed weight copy the NEXT eager forward (validation, inference) would reuse is stale now
        _conv.WEIGHT_EPOCH[0] += 1
        loss_dict, output_dict, bs = self.result
        if self.check_nan:
            assert not math.isnan(loss_dict[self.step.training_key].item()), "training_loss is NaN"
        return loss_dict, output_dict, bs


def make_adam(params, lr: float = 1e-4, weight_decay: float = 4e-4) -> torch.optim.Optimizer:
    return torch.optim.Adam(params, lr=lr, weight_decay=weight_decay)


class EvalStep:
    """EvaluationEpoch._step (runtime.py:345-383): move input*/target* tensors to the device, optional augmentation,
    forward under ``no_grad`` in eval mode -> (loss_dict, output_dict, batch_size).  ``save`` (optional, an args object
    with ``save`` and the ``save_result_*`` switches) makes every step also write its outputs like
    EvaluationEpoch.run does (runtime.py:423-424)."""

    def __init__(self, model_and_loss: ModelAndLoss, augmentation=None, device="cuda", save=None):
        self.mal, self.augmentation, self.device, self.save = model_and_loss, augmentation, device, save

    @torch.no_grad()
    def __call__(self, example_dict):
        self.mal.eval()
        for key, value in list(example_dict.items()):
            if ("input" in key or "target" in key) and torch.is_tensor(value):
                example_dict[key] = value.to(self.device)
        if self.augmentation is not None:
            example_dict = self.augmentation(example_dict)
        batch_size = example_dict["input1"].size(0)
        loss_dict, output_dict = self.mal(example_dict)
        if self.save is not None:
            from .io import save_outputs
            save_outputs(self.save, example_dict, output_dict)
        return loss_dict, output_dict, batch_size
