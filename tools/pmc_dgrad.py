"""Workload for profiles/r3_traffic_dgrad_128to565.txt: the 128 -> 565 data gradient of the context networks' first conv at
96x112x64 (five co-tile groups per pixel tile), three launches, for the FETCH_SIZE / WRITE_SIZE PMC passes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irr_amd import conv as C  # noqa: E402

B, H, W = 64, 96, 112
gy = torch.randn(B, 128, H, W, device="cuda")
w = torch.randn(128, 565, 3, 3, device="cuda") * 0.05
for _ in range(3):
    C.conv_dgrad(gy, w, 1, 1, (H, W))
torch.cuda.synchronize()
