"""Fused Adam for the IRR-PWC step: one kernel launch over flat parameter / gradient / moment arenas
(irr_adam_step_f32).  Same update rule and defaults as the reference's optimizer choice
(``torch.optim.Adam`` via optim/__init__.py:8-12 with lr=1e-4, weight_decay=4e-4,
scripts/IRR-PWC_flyingChairsOcc.sh:29-31).

``FusedAdam`` IS a ``torch.optim.Optimizer`` with one parameter group: the reference builds its learning-rate scheduler on the
optimizer object (configuration.py:579-608, ``MultiStepLR`` with milestones [54, 72, 90] and gamma 0.5 for FlyingChairsOcc,
scripts/IRR-PWC_flyingChairsOcc.sh:24-26) and schedulers read and write ``optimizer.param_groups[0]["lr"]``; the kernel's
hyper-parameters are taken from that group at every step."""
from __future__ import annotations

import torch

from . import conv, hip


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, module: torch.nn.Module, arena, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 4e-4, capturable: bool = False):
        """``arena`` is an irr_amd.ddp.GradArena built over ``module.named_parameters()``; parameters are
        re-homed into a flat arena with the same element order so a single launch updates everything."""
        if lr < 0.0 or eps < 0.0 or weight_decay < 0.0 or not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError(f"invalid Adam hyper-parameters: lr={lr} betas={betas} eps={eps} weight_decay={weight_decay}")
        self.arena = arena
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
        # ONE parameter group in arena order (the flat layout has one set of hyper-parameters per launch)
        super().__init__([p for _, p in arena.order],
                         dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))

    # the hyper-parameters live in param_groups[0] (what lr schedulers write); these are conveniences
    def _get(self, key):
        return self.param_groups[0][key]

    lr = property(lambda self: self._get("lr"), lambda self, v: self.param_groups[0].__setitem__("lr", v))
    betas = property(lambda self: self._get("betas"), lambda self, v: self.param_groups[0].__setitem__("betas", tuple(v)))
    eps = property(lambda self: self._get("eps"), lambda self, v: self.param_groups[0].__setitem__("eps", v))
    weight_decay = property(lambda self: self._get("weight_decay"),
                            lambda self, v: self.param_groups[0].__setitem__("weight_decay", v))

    def add_param_group(self, param_group):
        if getattr(self, "param_groups", None):
            raise ValueError("FusedAdam updates one flat arena with one set of hyper-parameters: a single parameter group")
        super().add_param_group(param_group)

    def hyper(self):
        """what a captured step baked into its kernel arguments (GraphedTrainStep re-captures when it changes)"""
        g = self.param_groups[0]
        return (float(g["lr"]), tuple(float(b) for b in g["betas"]), float(g["eps"]), float(g["weight_decay"]))

    def zero_grad(self, set_to_none: bool = False):
        """one memset of the flat gradient arena; the gradients stay views of it (``set_to_none`` is accepted for signature
        compatibility and ignored: a ``None`` gradient would detach the parameter from the arena)"""
        self.arena.zero_grad()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        # weight gradients routed past autograd (conv.WgradSide) are complete only after the lane's deferred folds: GradArena.sync()
        # does this in a data-parallel step; a plain ``loss.backward(); optimizer.step()`` loop gets it here
        side = getattr(self.arena, "_side_lane", None)
        if side is not None:
            side.join()
        self.t += 1                                  # (host copy: exact only outside graph replays)
        lr, (b1, b2), eps, wd = self.hyper()
        if self.capturable:
            self.step_dev += 1.0
        with hip.device_of(self.param_flat):
            hip.call("irr_adam_step_f32", hip.ptr(self.param_flat), hip.ptr(self.arena.flat), hip.ptr(self.exp_avg),
                     hip.ptr(self.exp_avg_sq), self.param_flat.numel(), lr, b1, b2, eps, wd,
                     1.0 - b1 ** self.t, 1.0 - b2 ** self.t, 1.0, hip.ptr(self.step_dev), hip.stream())
        # the kernel updates the parameters behind autograd's back: invalidate cached packed weights
        conv.WEIGHT_EPOCH[0] += 1
        return loss

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
        """flat moments + step count + the parameter group (the reference saves no optimizer state at all,
        configuration.py:192-314; this is the build's addition)"""
        t = int(self.step_dev.item()) if self.capturable else self.t
        g = self.param_groups[0]
        return {"t": t, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "param_group": {k: g[k] for k in ("lr", "betas", "eps", "weight_decay") if k in g} |
                               ({"initial_lr": g["initial_lr"]} if "initial_lr" in g else {})}

    @torch.no_grad()
    def load_state_dict(self, sd):
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.t = int(sd["t"])
        if self.step_dev is not None:
            self.step_dev.fill_(float(self.t))
        for k, v in sd.get("param_group", {}).items():
            self.param_groups[0][k] = tuple(v) if k == "betas" else v
