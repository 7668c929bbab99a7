"""Forward passes truncated after pyramid level K (PWCNet._debug_last_level), for a rocprofv3 --kernel-trace run:
   rocprofv3 --kernel-trace -d out -o t -- python3 tools/fwd_levels_trace.py 2 [B]
tools/rocpd_seq.py then lists the kernel sequence of the last pass (start, duration, gap to the previous kernel, blocks)."""
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import irr_amd  # noqa: E402
from irr_amd import ddp  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
args = types.SimpleNamespace(batch_size=B, model_div_flow=0.05)
torch.manual_seed(0)
model = irr_amd.PWCNet(args).cuda().train()
batch = bench.synthetic_batch(B, 384, 448, 1234, torch.device("cuda"))
model.__dict__["_debug_last_level"] = K
BWD = bool(os.environ.get("BWD"))          # BWD=1: forward + backward on ONE stream (weight gradients accumulated in the arena)
if BWD:
    arena = ddp.GradArena(model.named_parameters())
    arena.enable_direct_wgrad()
for it in range(5):
    with torch.no_grad() if os.environ.get("NOGRAD") else torch.enable_grad():
        out = model(batch)
    if BWD:
        arena.zero_grad()
        loss = sum(t.square().mean() for lv in out["flow"] for t in lv) + sum(t.square().mean() for lv in out["occ"] for t in lv)
        loss.backward()
        arena.sync()
    torch.cuda.synchronize()
    torch.cuda._sleep(2000000)          # marker between passes
    torch.cuda.synchronize()
