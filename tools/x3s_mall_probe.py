"""Does a chain of 32 -> 32 streaming convs run faster per sample when its tensors fit the 256 MB memory-side cache?
Ping-pong conv_forward between two (B, 32, H, W) buffers for several B; us per sample and launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irr_amd import conv as C  # noqa: E402

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (384, 448)
w = (torch.randn(32, 32, 3, 3) * 0.05).cuda()
b = torch.zeros(32).cuda()
for B in (2, 4, 8, 16, 32, 64):
    x = torch.randn(B, 32, H, W, device="cuda")
    y = torch.empty_like(x)
    r = torch.randn(B, 32, H, W, device="cuda")
    for mode in ("plain", "residual"):
        def step():
            C.conv_forward(x, w, b, 1, 1, True, out=y) if mode == "plain" else C.conv_forward(x, w, b, 1, 1, False, res=r, alpha=0.1, out=y)
            C.conv_forward(y, w, b, 1, 1, True, out=x) if mode == "plain" else C.conv_forward(y, w, b, 1, 1, False, res=r, alpha=0.1, out=x)
        for _ in range(3):
            step()
        n = max(4, 256 // B)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            step()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (2 * n)
        mb = B * 32 * H * W * 4 / 1e6
        print(f"B={B:3d} {mode:8s}: {us:8.1f} us/launch  {us / B:7.2f} us/sample   tensor {mb:7.1f} MB", flush=True)
    del x, y, r
