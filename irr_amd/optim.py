"""Fused Adam for the IRR-PWC step: one kernel launch over flat parameter / gradient / moment arenas
(irr_adam_step_f32).  Same update rule and defaults as the reference's optimizer choice
(``torch.optim.Adam`` via optim/__init__.py:8-12 with lr=1e-4, weight_decay=4e-4,
scripts/IRR-PWC_flyingChairsOcc.sh:29-31)."""
from __future__ import annotations

import torch

from . import conv, hip


class FusedAdam:
    def __init__(self, module: torch.nn.Module, arena, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 4e-4):
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

    def zero_grad(self, set_to_none: bool = False):
        self.arena.zero_grad()

    @torch.no_grad()
    def step(self):
        self.t += 1
        b1, b2 = self.betas
        with hip.device_of(self.param_flat):
            hip.call("irr_adam_step_f32", hip.ptr(self.param_flat), hip.ptr(self.arena.flat), hip.ptr(self.exp_avg),
                     hip.ptr(self.exp_avg_sq), self.param_flat.numel(), self.lr, b1, b2, self.eps, self.weight_decay,
                     1.0 - b1 ** self.t, 1.0 - b2 ** self.t, 1.0, hip.stream())
        # the kernel updates the parameters behind autograd's back: invalidate cached packed weights
        conv.WEIGHT_EPOCH[0] += 1

    def state_dict(self):
        return {"t": self.t, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq}
