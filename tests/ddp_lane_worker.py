"""Worker of tests/test_ddp_gpu.py (one process per GPU, started by torch.distributed.run): the REAL data-parallel path of
bench.py -- IRR-PWC + asynchronous weight-gradient lane + GradArena.sync() + RCCL -- against the single-process gradient
of the global batch, computed in the same process before the process group exists."""
import os
import sys
import types

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    backend = os.environ.get("IRR_DDP_BACKEND", "nccl")
    if backend != "nccl":                      # single-GPU boxes: both ranks share cuda:0 and exchange through gloo
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import irr_amd
    from irr_amd import ddp
    from irr_amd.train import ModelAndLoss
    from oracle import irr_pwc_oracle as O          # checker-side helpers only: synthetic weights and batch
    per, H, W = 2, 128, 192
    P = O.synthetic_params(0)
    full = {k: v.to(dev) for k, v in O.synthetic_batch(per * world, H, W, 1234).items()}

    def build(bs, reduce_fn):
        m = irr_amd.PWCNet(types.SimpleNamespace(batch_size=bs, model_div_flow=0.05), mask_threshold=0.9999)
        m.load_state_dict(P)
        m = m.to(dev).train()
        loss = irr_amd.MultiScaleEPE_PWC_Bi_Occ_upsample(types.SimpleNamespace(batch_size=bs, model_div_flow=0.05),
                                                         reduce_fn=reduce_fn).train()
        return m, ModelAndLoss(None, m, loss).train()

    # (a) single process, global batch (no process group yet -> the arena is a plain flat gradient buffer)
    m0, mal0 = build(per * world, None)
    a0 = ddp.GradArena(m0.named_parameters())
    a0.zero_grad()
    ld0, _ = mal0(full)
    ld0["total_loss"].backward()
    a0.sync()
    ref = a0.flat.clone()
    ref_loss = float(ld0["total_loss"].detach())
    # (b) one rank per GPU, lane + NCCL
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    m1, mal1 = build(per, ddp.reduce_losses())
    ddp.broadcast_params(m1)
    a1 = ddp.GradArena(m1.named_parameters())
    lane = os.environ.get("IRR_DDP_LANE", "async")      # async (the benched path) | inline (--no-async-wgrad) | plain (autograd hooks)
    if lane == "async":
        a1.enable_async_wgrad()
    elif lane == "inline":
        a1.enable_direct_wgrad()
    mine = ddp.shard_batch(full, rank, world)
    logs = []
    for _ in range(3):
        a1.zero_grad()
        ld1, _ = mal1(mine)
        ld1["total_loss"].backward()
        a1.sync()
        torch.cuda.synchronize()
        logs.append(list(a1.launch_log))
        if lane == "inline" and m1.branch_streams:
            # the shared-decoder bucket receives contributions from the main stream (flow branch) AND from the occlusion
            # branch's stream at the coarse levels; its all-reduce has to wait for both, whichever delivers the last one -- either
            # the arena saw both streams, or the lane folded launches of one stream on the other (and made it wait)
            assert len(a1._streams[1]) >= 2 or a1._side_lane.cross_stream_folds > 0, [len(s_) for s_ in a1._streams]
        err = (a1.flat - ref).double().norm().item() / ref.double().norm().item()
        assert err <= 1e-4, (rank, err)
        # total_loss is normalised by the per-rank batch; the mean over ranks is the global-batch loss
        t = ld1["total_loss"].detach().clone()
        dist.all_reduce(t)
        assert abs(float(t) / world - ref_loss) <= 2e-5 * abs(ref_loss), (float(t) / world, ref_loss)
    assert all(w == "sync" for _, w in logs[0]), logs                       # calibration step
    for lg in logs[1:]:                                                      # then: early + shared buckets start inside backward
        assert (0, "backward") in lg and (1, "backward") in lg, logs
    if lane != "plain":
        a1.disable_async_wgrad()
    if rank == 0:
        print("DDP_LANE_OK", logs[-1], flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
