import os, sys, torch
sys.path.insert(0, "/root/repo") if os.path.isdir("/root/repo") else None
sys.path.insert(0, os.getcwd())
from irr_amd import conv as C
B, H, W = 64, 96, 112
gy = torch.randn(B, 128, H, W, device="cuda"); w = torch.randn(128, 565, 3, 3, device="cuda") * 0.05
for _ in range(3):
    C.conv_dgrad(gy, w, 1, 1, (H, W))
torch.cuda.synchronize()
