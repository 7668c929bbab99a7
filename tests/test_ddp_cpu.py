"""world_size-2 gloo test (CPU) of the data-parallel pieces: batch sharding, gradient arena with bucketed
all-reduce, and the loss-scalar reduction that keeps flow/occ balancing identical to a single process."""
import os
import time
import types

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


class _LaneConv(torch.autograd.Function):
    """CPU stand-in for the asynchronous weight-gradient lane (irr_amd.conv.WgradSide): the weight / bias gradients are
    added straight into the arena views, autograd gets None, and the arena is told about the contribution."""

    @staticmethod
    def forward(ctx, x, w, b, arena_box):
        ctx.save_for_backward(x, w)
        ctx.objs = (w, b, arena_box)
        return torch.nn.functional.conv2d(x, w, b, padding=1)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        wobj, bobj, arena_box = ctx.objs
        with torch.no_grad():
            wobj.grad += torch.nn.grad.conv2d_weight(x, w.shape, gy, padding=1)
            bobj.grad += gy.sum(dim=(0, 2, 3))
        arena = arena_box[0]
        if arena is not None and arena.world > 1:
            arena._on_queue(wobj, bobj)          # (the real lane reports "queued" first, "folded into the gradient" later)
            arena._on_lane(wobj, bobj)
        gx = torch.nn.grad.conv2d_input(x.shape, w, gy, padding=1) if ctx.needs_input_grad[0] else None
        return gx, None, None, None


class Toy(nn.Module):
    def __init__(self):
        super().__init__()
        self.feature_pyramid_extractor = nn.Conv2d(3, 4, 3, padding=1)   # "late" bucket (back-propagated last)
        self.body = nn.Conv2d(4, 4, 3, padding=1)                         # "shared" bucket: used TWICE, gradients via the lane
        self.occ_shuffle_upsample = nn.Conv2d(4, 1, 3, padding=1)         # "early" bucket
        self.conv_1x1_1 = nn.Conv2d(4, 2, 1)
        self.arena_box = [None]

        self.stall = 0.0                                                  # seconds this rank sleeps in the middle of backward

    def forward(self, x):
        f = torch.tanh(self.feature_pyramid_extractor(x))
        h = torch.tanh(_LaneConv.apply(f, self.body.weight, self.body.bias, self.arena_box))
        if self.stall > 0 and h.requires_grad:
            h.register_hook(lambda g, t=self.stall: (time.sleep(t), g)[1])
        h = torch.tanh(_LaneConv.apply(h, self.body.weight, self.body.bias, self.arena_box))
        return self.conv_1x1_1(h), self.occ_shuffle_upsample(h)


def _loss(model, batch, reduce_fn):
    """CPU stand-in for the per-pixel terms (those are HIP kernels) + the product's balancing algebra."""
    from irr_amd.losses import balance_and_total
    flow, occ = model(batch["input1"])
    flow_loss = torch.norm(batch["target1"] - flow, dim=1).sum() + torch.norm(batch["target2"] - 0.5 * flow, dim=1).sum()
    s = torch.sigmoid(occ)
    occ_loss = -(batch["target_occ1"] * torch.log(s + 1e-8)).sum() - ((1 - batch["target_occ2"]) * torch.log(1 - s + 1e-8)).sum()
    return balance_and_total(flow_loss, occ_loss, batch["input1"].shape[0], reduce_fn)


def _make_batch(n):
    g = torch.Generator().manual_seed(11)
    return {"input1": torch.rand(n, 3, 8, 8, generator=g), "target1": torch.randn(n, 2, 8, 8, generator=g),
            "target2": torch.randn(n, 2, 8, 8, generator=g),
            "target_occ1": (torch.rand(n, 1, 8, 8, generator=g) < 0.3).float(),
            "target_occ2": (torch.rand(n, 1, 8, 8, generator=g) < 0.3).float()}


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from irr_amd import ddp
    torch.manual_seed(0)
    model = Toy()
    ddp.broadcast_params(model)
    arena = ddp.GradArena(model.named_parameters())
    model.arena_box[0] = arena
    assert [n for n, _ in arena.order] == ["occ_shuffle_upsample.weight", "occ_shuffle_upsample.bias", "conv_1x1_1.weight",
                                           "conv_1x1_1.bias", "body.weight", "body.bias",
                                           "feature_pyramid_extractor.weight", "feature_pyramid_extractor.bias"]
    full = _make_batch(8)
    mine = ddp.shard_batch(full, rank, world)
    logs = []
    for it in range(4):                      # calibration, then three rounds with the all-reduces started in backward;
        # in the last one ONE rank stalls 200 ms in the middle of backward (between the two uses of the shared conv): the other
        # ranks have started bucket 0 long before -- the collective ORDER is a property of the graph, not of the timing
        model.stall = 0.2 if (it == 3 and rank == world - 1) else 0.0
        arena.zero_grad()
        ld = _loss(model, mine, ddp.reduce_losses())
        ld["total_loss"].backward()
        arena.sync()
        logs.append(list(arena.launch_log))
    # step 1 learns the contribution counts (body: 2 lane contributions per step) and reduces everything at sync();
    # afterwards every bucket starts its all-reduce the moment its last contribution is in, in finalisation order
    assert logs[0] == [(0, "sync"), (1, "sync"), (2, "sync")], logs
    assert logs[1] == logs[2] == logs[3] == [(0, "backward"), (1, "backward"), (2, "backward")], logs
    assert arena._expected_queued == [0, 2, 0], arena._expected_queued
    assert len(arena.launch_times) == 3 and arena.launch_times == sorted(arena.launch_times)
    if rank == world - 1:
        assert arena.launch_times[1] >= 200.0, arena.launch_times          # bucket 1 completes after the stall
    exp = {n: arena._expected[id(p)] for n, p in model.named_parameters()}
    # (2 lane contributions; torch additionally runs the AccumulateGrad hook once for a parameter whose Function returned None)
    assert exp["body.weight"] in (2, 3) and exp["body.bias"] in (2, 3) and exp["conv_1x1_1.weight"] == 1, exp
    # plain lists, not tensors: a tensor travels by file-descriptor passing and is lost if this process exits first
    q.put((rank, arena.flat.tolist(), float(ld["total_loss"].detach())))
    dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("world", [2, 4])
def test_multi_rank_gloo_matches_single_process(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 13 * world) % 1000
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=150) for _ in range(world)], key=lambda t: t[0])
    res = [(r, torch.tensor(flat), loss) for r, flat, loss in res]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for r in range(1, world):
        assert torch.allclose(res[0][1], res[r][1])                   # identical after all-reduce
    # single process on the global batch: per-rank loss divides by the PER-RANK batch (losses.py:569-571),
    # so mean over ranks of grads == grads of (global loss / global batch) * ... check against that
    from irr_amd import ddp
    torch.manual_seed(0)
    model = Toy()
    arena = ddp.GradArena(model.named_parameters())
    model.arena_box[0] = arena
    full = _make_batch(8)
    arena.zero_grad()
    ld = _loss(model, full, None)
    ld["total_loss"].backward()
    assert torch.allclose(arena.flat, res[0][1], rtol=1e-4, atol=1e-6), (arena.flat - res[0][1]).abs().max()


def _failsafe_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import warnings
    from irr_amd import ddp
    torch.manual_seed(0)
    model = Toy()
    ddp.broadcast_params(model)
    arena = ddp.GradArena(model.named_parameters())
    model.arena_box[0] = arena
    mine = ddp.shard_batch(_make_batch(4), rank, world)
    out = {}
    # (a) the ranks' calibration steps see DIFFERENT contribution counts (rank 1 reports one extra lane contribution): the arena
    #     must notice (one MIN / MAX all-reduce), stay in reduce-at-sync mode on EVERY rank for that step and calibrate on the next
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for it in range(2):
            arena.zero_grad()
            ld = _loss(model, mine, ddp.reduce_losses())
            ld["total_loss"].backward()
            if rank == 1 and it == 0:
                arena._on_lane(model.body.weight, None)
            arena.sync()
            out[f"log{it}"] = list(arena.launch_log)
    out["mismatch"] = arena.calibration_mismatch
    out["calibrated_after_mismatch"] = arena._expected is not None
    # (b) calibrated, then ONE rank delivers a contribution after its bucket's all-reduce was started: no exception inside the
    #     hook (the other rank would hang in its matching collective); sync() completes the step's collectives, then raises there
    arena.recalibrate()
    arena.zero_grad()
    _loss(model, mine, ddp.reduce_losses())["total_loss"].backward()
    arena.sync()
    assert arena._expected is not None
    arena.zero_grad()
    _loss(model, mine, ddp.reduce_losses())["total_loss"].backward()
    late = None
    if rank == 1:
        arena._on_lane(model.occ_shuffle_upsample.weight, None)        # bucket 0 was launched inside backward already
    try:
        arena.sync()
    except RuntimeError as e:
        late = str(e)[:40]
    out["late"] = late
    # (c) the rank that saw the late contribution stays refused (it keeps its calibrated schedule, so the collectives of the other
    #     rank's next step still find their partners) until recalibrate() is called on every rank; then training goes on
    arena.zero_grad()
    _loss(model, mine, ddp.reduce_losses())["total_loss"].backward()
    again = None
    try:
        arena.sync()
    except RuntimeError as e:
        again = str(e)[:40]
    out["again"] = again
    out["log_after_late"] = list(arena.launch_log)
    arena.recalibrate()
    for it in range(2):
        arena.zero_grad()
        _loss(model, mine, ddp.reduce_losses())["total_loss"].backward()
        arena.sync()
        out[f"relog{it}"] = list(arena.launch_log)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_arena_is_fail_safe_when_ranks_disagree():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 7) % 1000
    procs = [ctx.Process(target=_failsafe_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=100) for _ in range(2))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for r in (0, 1):
        # the step after the mismatch still reduces at sync() and is the new calibration step (the ranks agree there)
        assert res[r]["mismatch"] is True and res[r]["calibrated_after_mismatch"] is True, res
        assert res[r]["log0"] == res[r]["log1"] == [(0, "sync"), (1, "sync"), (2, "sync")], res
    assert res[0]["late"] is None and res[1]["late"] is not None and "contribution" in res[1]["late"], res
    assert res[0]["again"] is None and res[1]["again"] == res[1]["late"], res
    for r in (0, 1):
        assert res[r]["log_after_late"] == [(0, "backward"), (1, "backward"), (2, "backward")], res
        assert res[r]["relog0"] == [(0, "sync"), (1, "sync"), (2, "sync")], res
        assert res[r]["relog1"] == [(0, "backward"), (1, "backward"), (2, "backward")], res


def test_shard_batch():
    from irr_amd import ddp
    b = {"input1": torch.arange(8).view(8, 1), "index": 3}
    s = ddp.shard_batch(b, 1, 4)
    assert s["input1"].flatten().tolist() == [2, 3] and s["index"] == 3
