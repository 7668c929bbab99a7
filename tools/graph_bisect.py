"""Root-causing the drift / NaN of hipGraph replays captured WITHOUT the asynchronous weight-gradient lane.
Runs the SAME N steps (same batches) eagerly and as replays for one configuration and prints the parameter difference:

    LANE=none|direct|async [CONTROL=1] [IRR_WARP_BWD_ATOMIC=1] [IRR_ZERO_MEMSET=1]  python tools/graph_bisect.py [B H W]

CONTROL=1 also runs the eager steps twice (what two runs of the SAME configuration differ by: atomics + Adam's sign amplification).
Result (profiles/r4_graph_bisect.txt): the drift needs BOTH the atomic warp backward (IRR_WARP_BWD_ATOMIC=1) and its zero fill as
hipMemsetAsync (IRR_ZERO_MEMSET=1) -- a graph memset node is not ordered against the kernels around it.
"""
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import irr_amd  # noqa: E402
from irr_amd import ddp  # noqa: E402
from irr_amd.optim import FusedAdam  # noqa: E402
from irr_amd.train import GraphedTrainStep, ModelAndLoss, TrainStep  # noqa: E402

B, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (2, 128, 192)
N = int(os.environ.get("STEPS", 6))
lane = os.environ.get("LANE", "direct")
batches = [bench.synthetic_batch(B, H, W, 100 + i, torch.device("cuda")) for i in range(3)]


def run(graph):
    args = types.SimpleNamespace(batch_size=B, model_div_flow=0.05)
    torch.manual_seed(0)
    model = irr_amd.PWCNet(args).cuda().train()
    arena = ddp.GradArena(model.named_parameters())
    if lane == "direct":
        arena.enable_direct_wgrad()
    elif lane == "async":
        arena.enable_async_wgrad()
    opt = FusedAdam(model, arena, capturable=graph)
    step = TrainStep(ModelAndLoss(args, model, irr_amd.MultiScaleEPE_PWC_Bi_Occ_upsample(args).train()), opt, grad_sync=arena.sync)
    if graph:
        step = GraphedTrainStep(step)
    losses, grads = [], []
    for i in range(N):
        ld, _, _ = step({k: v.clone() for k, v in batches[i % 3].items()})
        losses.append(float(ld["total_loss"].detach()))
        grads.append(arena.flat.double().norm().item())
    torch.cuda.synchronize()
    p = torch.cat([q.detach().reshape(-1) for q in model.parameters()]).double().clone()
    arena.disable_async_wgrad()
    return losses, grads, p


le, ge, pe = run(False)
if os.environ.get("CONTROL"):                  # eager vs eager: what two runs of the SAME configuration differ by (atomics + Adam)
    lc, gc, pc = run(False)
    print(f"LANE={lane} {B}x{H}x{W}: parameters after {N} steps, eager vs eager (control): {(pc - pe).norm().item() / pe.norm().item():.3e}")
lg, gg, pg = run(True)
d = (pg - pe).norm().item() / pe.norm().item()
print(f"LANE={lane} WARP_BWD_ATOMIC={os.environ.get('IRR_WARP_BWD_ATOMIC', '0')} ZERO_MEMSET={os.environ.get('IRR_ZERO_MEMSET', '0')} "
      f"{B}x{H}x{W}: parameters after {N} steps, replay vs eager: {d:.3e}")
print("  eager  losses", [f"{v:.5f}" for v in le], "grad norms", [f"{v:.4f}" for v in ge])
print("  replay losses", [f"{v:.5f}" for v in lg], "grad norms", [f"{v:.4f}" for v in gg], flush=True)
