"""GPU time of one IRR-PWC train step by pyramid level: forward per level (events at the level boundaries of PWCNet.forward,
recorded through a forward pre-hook on the first module each level calls) is hard to hook from outside, so this tool times
passes truncated after level k (PWCNet._debug_last_level) and differences them; forward only and forward + backward."""
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import irr_amd  # noqa: E402
from irr_amd import ddp  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
args = types.SimpleNamespace(batch_size=B, model_div_flow=0.05)
torch.manual_seed(0)
model = irr_amd.PWCNet(args).cuda().train()
arena = ddp.GradArena(model.named_parameters())
if "--lane" in sys.argv:
    arena.enable_async_wgrad()
batch = bench.synthetic_batch(B, 384, 448, 1234, torch.device("cuda"))
full = model.num_levels


def run(levels, backward):
    model.__dict__["_debug_last_level"] = levels - 1
    ts = []
    for it in range(4):
        arena.zero_grad()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = model(batch)
        if backward:
            loss = sum(t.square().mean() for lv in out["flow"] for t in lv) + sum(t.square().mean() for lv in out["occ"] for t in lv)
            loss.backward()
            arena.sync()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return min(ts[1:]) * 1e3


prev_f = prev_b = 0.0
print("levels  fwd_ms  (+level)   fwd+bwd_ms  (+level)")
for lv in range(1, full + 1):
    f = run(lv, False)
    b = run(lv, True)
    print(f"0..{lv - 1}   {f:7.2f}  {f - prev_f:7.2f}     {b:8.2f}  {b - prev_b:8.2f}")
    prev_f, prev_b = f, b
