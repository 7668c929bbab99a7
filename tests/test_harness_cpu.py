"""CPU-only: the host logic of the drop-in route (irr_amd/harness.py, GradArena.adopt_grads, lane views) -- no kernels run."""
import types

import pytest
import torch
import torch.nn as nn


def _toy():
    torch.manual_seed(0)
    return nn.Sequential(nn.Conv2d(3, 4, 3), nn.Conv2d(4, 2, 3))


def test_adopt_grads_restores_arena_views_after_any_zero_grad():
    """optimizer.zero_grad() of a stock optimizer sets the gradients to None (torch's default) or zeroes them in place; either way the
    next forward pass must find every .grad a zeroed view of the flat arena, accumulation across backward passes must survive, and a
    foreign gradient tensor must be carried over."""
    from irr_amd import ddp
    m = _toy()
    arena = ddp.GradArena(m.named_parameters())
    params = list(m.parameters())
    assert all(arena._inside(p.grad) for p in params)
    # (1) set_to_none
    arena.flat.fill_(3.0)
    for p in params:
        p.grad = None
    arena.adopt_grads()
    assert all(p.grad is not None and arena._inside(p.grad) and float(p.grad.abs().sum()) == 0.0 for p in params)
    # (2) accumulated gradients that still are the arena's views are kept as they are
    arena.flat.fill_(2.0)
    arena.adopt_grads()
    assert float(arena.flat.min()) == 2.0
    # (3) one gradient cleared, one replaced by a foreign tensor
    params[0].grad = None
    params[1].grad = torch.full_like(params[1], 7.0)
    arena.adopt_grads()
    assert arena._inside(params[0].grad) and float(params[0].grad.abs().sum()) == 0.0
    assert arena._inside(params[1].grad) and float(params[1].grad.min()) == 7.0
    assert float(params[2].grad.min()) == 2.0
    # the views tile the arena in parameter order
    off = 0
    for _, p in arena.order:
        assert p.grad.data_ptr() == arena.flat.data_ptr() + 4 * off
        off += p.numel()


def test_lane_routes_by_identity_not_by_id():
    """The lane looks gradients up by id(parameter) and verifies the identity: a parameter of ANOTHER model must never be routed into
    this model's arena, even if a dead parameter's id is reused (round 4: ten gradients of a test model landed in the arena of a model
    that had been garbage-collected)."""
    from irr_amd import conv as C, ddp
    m = _toy()
    arena = ddp.GradArena(m.named_parameters())
    lane = C.WgradSide([(p, p.grad) for _, p in arena.order], inline=True)
    w, b = m[0].weight, m[0].bias
    gw, gb = lane.route(w, b)
    assert gw.data_ptr() == w.grad.data_ptr() and gb.data_ptr() == b.grad.data_ptr()
    other = _toy()
    assert lane.route(other[0].weight, other[0].bias) is None
    # simulate id reuse: an entry whose weak reference points at a different (or dead) object
    import weakref
    lane.views[id(other[0].weight)] = (weakref.ref(w), gw)
    assert lane.route(other[0].weight, None) is None


def test_auto_install_is_a_noop_without_cuda_parameters_or_in_eval(monkeypatch):
    from irr_amd import conv as C, harness
    m = _toy().train()
    assert C.SIDE is None
    harness.auto_install(m)                      # CPU parameters: nothing to install (the kernels would raise anyway)
    assert C.SIDE is None and not harness.installed(m)
    m.eval()
    harness.auto_install(m)
    assert C.SIDE is None
    harness.set_enabled(False)
    try:
        assert not harness.enabled()
        harness.auto_install(m.train())
        assert C.SIDE is None
    finally:
        harness.set_enabled(True)


def test_fused_adam_is_a_torch_optimizer_class():
    """(construction needs the GPU; the class relationship and the single-group rule do not)"""
    from irr_amd.optim import FusedAdam
    assert issubclass(FusedAdam, torch.optim.Optimizer)
    for name in ("lr", "betas", "eps", "weight_decay"):
        assert isinstance(getattr(FusedAdam, name), property)


def test_kernel_timer_samples_steps():
    from irr_amd import conv as C
    t = C.KernelTimer()
    calls = []
    t.begin_step(False)
    t.wrap(1, 1.0, lambda: calls.append("a"))    # inactive step: the launch runs, no events are created (no GPU needed)
    assert calls == ["a"] and t.records == [] and t.steps == 0


def test_gradients_cleared_between_forward_and_backward_are_reattached():
    """``out = model(x); opt.zero_grad(); loss.backward(); opt.step()`` -- torch's zero_grad sets ``.grad`` to None
    AFTER the forward pass adopted the arena views; the lane still accumulates into the arena slices and hands autograd None, so
    without the end-of-backward re-adoption a stock optimizer would skip the parameter.  Host logic only (inline lane, CPU tensors)."""
    from irr_amd import conv as C, ddp
    m = _toy()
    arena = ddp.GradArena(m.named_parameters())
    lane = C.WgradSide([(p, p.grad) for _, p in arena.order], inline=True)
    lane.batch = None
    lane.on_join = arena.readopt_routed
    w, b = m[0].weight, m[0].bias
    w2 = m[1].weight
    arena.flat.fill_(9.0)                        # the previous step's gradients are still in the arena ...
    for p in m.parameters():                     # ... when the caller's zero_grad() comes after the forward pass
        p.grad = None
    gw, gb = lane.route(w, b)

    def fake_wgrad():                            # what a routed weight-gradient launch does: accumulate into the arena slices
        gw.add_(2.0)
        gb.add_(3.0)
    lane.launch(fake_wgrad, (), (w, b), gw=None)
    w2.grad = torch.full_like(w2, 5.0)           # a gradient autograd delivered itself (a contribution that did not take the lane)
    g2, _ = lane.route(w2, None)
    lane.launch(lambda: g2.add_(1.0), (), (w2, None), gw=None)
    assert w.grad is None
    lane.join()
    assert w.grad is not None and arena._inside(w.grad) and float(w.grad.min()) == 2.0 and float(b.grad.min()) == 3.0
    assert arena._inside(w2.grad) and float(w2.grad.min()) == 6.0          # routed 1.0 + foreign 5.0, now one arena view
    assert m[1].bias.grad is None                # never routed in this pass: left alone
    lane.join()                                  # nothing routed since: a no-op
    assert float(w.grad.min()) == 2.0


def test_lane_is_cleaned_up_after_a_backward_pass_that_raised():
    """autograd skips its final callbacks when backward raises, so ``_join_queued`` stayed True for ever (no later
    pass queued its join) and the failed pass's queued launches / fold jobs leaked into the next step.  ``stale()`` / ``abandon()``
    are what irr_amd.harness calls at the start of the next training forward pass."""
    from irr_amd import conv as C, ddp
    m = _toy()
    arena = ddp.GradArena(m.named_parameters())
    lane = C.WgradSide([(p, p.grad) for _, p in arena.order], inline=True)
    lane.batch = None
    assert not lane.stale()
    lane._join_queued = True                     # a backward pass queued its join, then raised before the callback ran
    lane._pending.append((m[0].weight, None))
    lane._queued.append((lambda: None, (), (m[0].weight, None), 0))
    assert lane.stale()
    lane.abandon()
    assert not lane.stale() and lane._queued == [] and lane._pending == [] and not lane._join_queued
