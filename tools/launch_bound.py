"""Is the forward pass launch-bound at the coarse pyramid levels?  CPU issue time vs GPU time between the per-level
cost-volume calls of one IRR-PWC train step."""
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import irr_amd  # noqa: E402
from irr_amd import ddp, functional as Fn  # noqa: E402
from irr_amd.optim import FusedAdam  # noqa: E402
from irr_amd.train import ModelAndLoss, TrainStep  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
args = types.SimpleNamespace(batch_size=B, model_div_flow=0.05)
torch.manual_seed(0)
model = irr_amd.PWCNet(args).cuda().train()
loss = irr_amd.MultiScaleEPE_PWC_Bi_Occ_upsample(args).train()
arena = ddp.GradArena(model.named_parameters())
arena.enable_async_wgrad()
step = TrainStep(ModelAndLoss(args, model, loss), FusedAdam(model, arena), grad_sync=arena.sync)
batch = bench.synthetic_batch(B, 384, 448, 1234, torch.device("cuda"))
for _ in range(3):
    step(batch)
torch.cuda.synchronize()
marks = []


def mark(name):
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    marks.append((name, time.perf_counter(), ev))


orig_cv = Fn.cost_volume


def cv(*a, **k):
    mark("level %dx%d" % tuple(a[0].shape[2:]))
    return orig_cv(*a, **k)


Fn.cost_volume = cv
orig_bw = torch.Tensor.backward


def bw(self, *a, **k):
    mark("backward")
    r = orig_bw(self, *a, **k)
    mark("backward issued")
    return r


torch.Tensor.backward = bw
mark("step start")
step(batch)
mark("step issued")
torch.cuda.synchronize()
t_end = time.perf_counter()
print(f"{'mark':18s} {'cpu ms':>8s} {'gpu ms':>8s}   (time since the previous mark; gpu ~= cpu and small kernels => launch-bound)")
for (n0, c0, e0), (n1, c1, e1) in zip(marks[:-1], marks[1:]):
    print(f"{n0:18s} {1e3 * (c1 - c0):8.2f} {e0.elapsed_time(e1):8.2f}")
print(f"step: cpu issue {1e3 * (marks[-1][1] - marks[0][1]):.1f} ms, wall {1e3 * (t_end - marks[0][1]):.1f} ms")
