"""Trajectory check: N optimisation steps from the same initialisation with IRR_CONV_MATH = x3 and f32 (asynchronous
wgrad lane on, FusedAdam): total_loss per step, and the parameter drift between the two runs."""
import os, sys, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import irr_amd, bench
from irr_amd import conv as C, ddp
from irr_amd.optim import FusedAdam
from irr_amd.train import ModelAndLoss, TrainStep

N = int(os.environ.get("SOAK_STEPS", 12))
B, H, W = int(os.environ.get("SOAK_B", 8)), 384, 448


def run(math):
    C.set_math(math)
    args = types.SimpleNamespace(batch_size=B, model_div_flow=0.05)
    torch.manual_seed(0)
    model = irr_amd.PWCNet(args).cuda().train()
    loss = irr_amd.MultiScaleEPE_PWC_Bi_Occ_upsample(args).train()
    arena = ddp.GradArena(model.named_parameters())
    arena.enable_async_wgrad()
    step = TrainStep(ModelAndLoss(args, model, loss), FusedAdam(model, arena, lr=1e-4, weight_decay=4e-4), grad_sync=arena.sync)
    losses = []
    for i in range(N):
        batch = bench.synthetic_batch(B, H, W, 1234 + i, torch.device("cuda"))
        ld, _, _ = step(batch)
        losses.append(float(ld["total_loss"].detach()))
    arena.disable_async_wgrad()
    return losses, torch.cat([p.detach().flatten() for p in model.parameters()]).clone()


la, pa = run("x3")
lb, pb = run("f32")
for i, (a, b) in enumerate(zip(la, lb)):
    print(f"step {i:2d}: total_loss x3 {a:.6f}  f32 {b:.6f}  rel diff {abs(a - b) / abs(b):.2e}")
print("parameter drift |x3 - f32| / |f32| =", float((pa - pb).norm() / pb.norm()))
assert all(l == l for l in la), "NaN"
