"""GPU KERNEL time of the coarse pyramid levels of one IRR-PWC train step (round 5).

tools/level_times.py times truncated passes with a synchronisation around each: at levels 0-2 such a pass is shorter than its own
host issue, so that number is the HOST's, while in the real step the host runs a whole backward ahead (no sync between steps) and
these levels cost their kernel time.  This tool runs passes truncated after level L back to back (no sync inside) -- run it under
`rocprofv3 --kernel-trace` and read the kernel table (tools/rocpd_stats.py): total kernel time / PASSES = GPU time of levels 0..L.

usage: python3 tools/levels_kernel_time.py <last level> [batch]"""
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import irr_amd  # noqa: E402
from irr_amd import ddp  # noqa: E402

LAST = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
PASSES = 10
args = types.SimpleNamespace(batch_size=B, model_div_flow=0.05)
torch.manual_seed(0)
model = irr_amd.PWCNet(args).cuda().train()
arena = ddp.GradArena(model.named_parameters())
batch = bench.synthetic_batch(B, 384, 448, 1234, torch.device("cuda"))
model.__dict__["_debug_last_level"] = LAST


def one():
    arena.zero_grad()
    out = model(batch)
    loss = sum(t.square().mean() for lv in out["flow"] for t in lv) + sum(t.square().mean() for lv in out["occ"] for t in lv)
    loss.backward()
    arena.sync()


one()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(PASSES - 1):
    one()
host = time.perf_counter() - t0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(f"levels 0..{LAST}  batch {B}: {PASSES} passes in the trace; back to back {wall / (PASSES - 1) * 1e3:.2f} ms / pass wall, "
      f"{host / (PASSES - 1) * 1e3:.2f} ms / pass host issue")
