"""How long does the HOST need to issue one train step (async launches), vs the GPU time of the step?"""
import os, sys, time, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import irr_amd, bench
from irr_amd import ddp
from irr_amd.optim import FusedAdam
from irr_amd.train import ModelAndLoss, TrainStep
B = 32
args = types.SimpleNamespace(batch_size=B, model_div_flow=0.05)
torch.manual_seed(0)
model = irr_amd.PWCNet(args).cuda().train()
loss = irr_amd.MultiScaleEPE_PWC_Bi_Occ_upsample(args).train()
arena = ddp.GradArena(model.named_parameters())
if "--serial" not in sys.argv:
    arena.enable_async_wgrad()
step = TrainStep(ModelAndLoss(args, model, loss), FusedAdam(model, arena), grad_sync=arena.sync, check_nan="--nonan" not in sys.argv)
batch = bench.synthetic_batch(B, 384, 448, 1234, torch.device("cuda"))
for _ in range(2):
    step(batch)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    step(batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host issue {1e3*(t1-t0):.1f} ms, step wall {1e3*(t2-t0):.1f} ms")
