"""Eager vs hipGraph-replayed IRR-PWC train step (bench.py's configuration): ms/step and a parity check of the two."""
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import irr_amd  # noqa: E402
from irr_amd import ddp  # noqa: E402
from irr_amd.optim import FusedAdam  # noqa: E402
from irr_amd.train import GraphedTrainStep, ModelAndLoss, TrainStep  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (384, 448)
steps = int(os.environ.get('STEPS', 8))


def make(graph):
    args = types.SimpleNamespace(batch_size=B, model_div_flow=0.05)
    torch.manual_seed(0)
    model = irr_amd.PWCNet(args).cuda().train()
    loss = irr_amd.MultiScaleEPE_PWC_Bi_Occ_upsample(args).train()
    arena = ddp.GradArena(model.named_parameters())
    if os.environ.get("LANE", "async") == "direct":        # single stream, gradients accumulated straight into the arena
        arena.enable_direct_wgrad()
    elif os.environ.get("LANE", "async") != "none":          # "none": weight gradients through autograd
        arena.enable_async_wgrad()
    step = TrainStep(ModelAndLoss(args, model, loss), FusedAdam(model, arena, capturable=graph), grad_sync=arena.sync)
    return model, arena, (GraphedTrainStep(step) if graph else step)


batch = bench.synthetic_batch(B, H, W, 1234, torch.device("cuda"))
res = {}
for graph in ((True,) if os.environ.get("GRAPH_ONLY") else (False, True)):
    model, arena, step = make(graph)
    losses = []
    for _ in range(3):
        ld, _, _ = step(batch)
        losses.append(float(ld["total_loss"].detach()))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ld, _, _ = step(batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    losses.append(float(ld["total_loss"].detach()))
    res[graph] = (dt, losses, torch.cat([p.detach().reshape(-1) for p in model.parameters()]).double())
    print(f"{'hipGraph replay' if graph else 'eager          '}: {dt * 1e3:8.2f} ms/step  {B / dt:7.2f} pairs/s  losses {losses}", flush=True)
    arena.disable_async_wgrad()
    del model, arena, step
    if os.environ.get("GC"):
        import gc
        gc.collect()
    if not os.environ.get("NO_EMPTY"):
        torch.cuda.empty_cache()
if False in res:
    d = (res[True][2] - res[False][2]).norm().item() / res[False][2].norm().item()
    print(f"parameters after {3 + steps} steps, graph vs eager: relative difference {d:.2e}")
