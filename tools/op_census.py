"""Which torch (non-irr_amd) device ops does one IRR-PWC train step launch, and from which source line?
torch.profiler with stacks; prints per (op, python frame inside the repo) the launch count and device time."""
import collections
import os
import sys
import types

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import irr_amd  # noqa: E402
from irr_amd import ddp  # noqa: E402
from irr_amd.optim import FusedAdam  # noqa: E402
from irr_amd.train import ModelAndLoss, TrainStep  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
args = types.SimpleNamespace(batch_size=B, model_div_flow=0.05)
torch.manual_seed(0)
model = irr_amd.PWCNet(args).cuda().train()
loss = irr_amd.MultiScaleEPE_PWC_Bi_Occ_upsample(args).train()
arena = ddp.GradArena(model.named_parameters())
if "--serial" not in sys.argv:
    arena.enable_async_wgrad()
step = TrainStep(ModelAndLoss(args, model, loss), FusedAdam(model, arena), grad_sync=arena.sync)
batch = bench.synthetic_batch(B, 384, 448, 1234, torch.device("cuda"))
for _ in range(2):
    step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step(batch)
    torch.cuda.synchronize()

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
agg = collections.defaultdict(lambda: [0, 0.0])
tot = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_time_total <= 0 or not ev.name.startswith("aten::") or ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::"):
        continue
    frame = "?"
    for fr in ev.stack or []:
        if root in fr and "tools/op_census" not in fr:
            frame = fr.replace(root + "/", "")
            break
    if frame == "?":
        frame = "autograd engine" if not ev.stack else (ev.stack[0][:60])
    shapes = str(ev.input_shapes)[:60]
    a = agg[(ev.name, frame, shapes)]
    a[0] += 1
    a[1] += ev.device_time_total
    t = tot[ev.name]
    t[0] += 1
    t[1] += ev.device_time_total
# launches by the outermost enclosing profiler range (autograd node in backward, module-level op in forward)
top = collections.defaultdict(lambda: [0, 0.0])
acc = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_time_total <= 0 or not ev.name.startswith("aten::") or (ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::")):
        continue
    anc = ev
    while anc.cpu_parent is not None:
        anc = anc.cpu_parent
    key = (anc.name[:70], ev.name)
    top[key][0] += 1
    top[key][1] += ev.device_time_total
    if ev.name in ("aten::add_", "aten::add"):                 # the engine's gradient accumulations: which node's outputs, which shapes
        k2 = (anc.name.replace("autograd::engine::evaluate_function: ", "")[:40], str(ev.input_shapes[0])[:30])
        acc[k2][0] += 1
        acc[k2][1] += ev.device_time_total
print("== torch ops by outermost range (autograd node) ==")
for k, v in sorted(top.items(), key=lambda kv: -kv[1][0])[:45]:
    print(f"{v[0]:5d} {v[1] / 1e3:7.2f} ms  {k[0]:70s} {k[1]}")
print("== gradient accumulations (aten::add_ / add) by producing node and shape ==")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{v[0]:5d} {v[1] / 1e3:7.2f} ms  {k[0]:40s} {k[1]}")
print("== per op ==")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{k:40s} {v[0]:5d} launches {v[1] / 1e3:8.2f} ms")
print("== per (op, frame, shapes) ==")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    print(f"{k[0]:28s} {v[0]:4d} {v[1] / 1e3:7.2f} ms  {k[1]}  {k[2]}")
