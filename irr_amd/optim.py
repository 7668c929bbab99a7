"""Fused Adam for the IRR-PWC step: one kernel launch over flat parameter / gradient / moment arenas
(irr_adam_step_f32).  Same update rule and defaults as the reference's optimizer choice
(``torch.optim.Adam`` via optim/__init__.py:8-12 with lr=1e-4, weight_decay=4e-4,
scripts/IRR-PWC_flyingChairsOcc.sh:29-31)."""
from __future__ import annotations

import torch

from . import conv, hip


class FusedAdam:
    def __init__(self, module: torch.nn.Module, arena, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 4e-4, capturable: bool = False):
        """``arena`` is an irr_amd.ddp.GradArena built over ``module.named_parameters()``; parameters are
        re-homed into a flat arena with the same element order so a single launch updates everything."""
        self.arena = arena
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        n = arena.flat.numel()
        dev = arena.flat.device
        self.param_flat = torch.empty(n, device=dev, dtype=torch.float32)
        off = 0
        with torch.no_grad():
            for _, p in arena.order:
                k = p.numel()
                self.param_flat[off:off + k].copy_(p.detach().reshape(-1))
                p.data = self.param_flat[off:off + k].view_as(p)
                off += k
        self.exp_avg = torch.zeros(n, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(n, device=dev, dtype=torch.float32)
        self.t = 0
        # capturable: the step count lives in a device scalar that the kernel reads, so a hipGraph-captured step stays correct
        # when replayed (torch.optim.Adam(capturable=True) has the same meaning)
        self.capturable = capturable
        self.step_dev = torch.zeros(1, device=dev, dtype=torch.float32) if capturable else None

    def hyper(self):
        """what a captured step baked into its kernel arguments (GraphedTrainStep re-captures when it changes)"""
        return (float(self.lr), tuple(float(b) for b in self.betas), float(self.eps), float(self.weight_decay))

    def zero_grad(self, set_to_none: bool = False):
        self.arena.zero_grad()

    @torch.no_grad()
    def step(self):
        self.t += 1                                  # (host copy: exact only outside graph replays)
        b1, b2 = self.betas
        if self.capturable:
            self.step_dev += 1.0
        with hip.device_of(self.param_flat):
            hip.call("irr_adam_step_f32", hip.ptr(self.param_flat), hip.ptr(self.arena.flat), hip.ptr(self.exp_avg),
                     hip.ptr(self.exp_avg_sq), self.param_flat.numel(), self.lr, b1, b2, self.eps, self.weight_decay,
                     1.0 - b1 ** self.t, 1.0 - b2 ** self.t, 1.0, hip.ptr(self.step_dev), hip.stream())
        # the kernel updates the parameters behind autograd's back: invalidate cached packed weights
        conv.WEIGHT_EPOCH[0] += 1

    def snapshot(self):
        """copies of everything a step changes (parameters, moments, step count) -- see restore()"""
        return (self.param_flat.clone(), self.exp_avg.clone(), self.exp_avg_sq.clone(), self.t,
                self.step_dev.clone() if self.step_dev is not None else None)

    @torch.no_grad()
    def restore(self, snap):
        self.param_flat.copy_(snap[0])
        self.exp_avg.copy_(snap[1])
        self.exp_avg_sq.copy_(snap[2])
        self.t = snap[3]
        if self.step_dev is not None:
            self.step_dev.copy_(snap[4])
        conv.WEIGHT_EPOCH[0] += 1

    def state_dict(self):
        t = int(self.step_dev.item()) if self.capturable else self.t
        return {"t": t, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq}
