"""Workload for the PMC passes of profiles/r1e_pmc_*.txt: the heaviest conv shapes on the x3 kernels and the cost-volume /
warp kernels at the 96x112 level (2B = 64 samples), three launches each."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irr_amd import conv as C, functional as Fn
B, H, W = 64, 96, 112
x = torch.randn(B, 565, H, W, device="cuda"); w = torch.randn(128, 565, 3, 3, device="cuda") * 0.05; b = torch.randn(128, device="cuda")
gy = torch.randn(B, 128, H, W, device="cuda"); gw = torch.zeros(128, 565, 3, 3, device="cuda")
f1 = torch.randn(B, 32, H, W, device="cuda", requires_grad=True); f2 = torch.randn(B, 32, H, W, device="cuda", requires_grad=True)
fl = (torch.randn(B, 2, H, W, device="cuda") * 0.02).requires_grad_(True)
for _ in range(3):
    C.conv_forward(x, w, b, 1, 1, True)
    C.conv_wgrad(x, gy, (128, 565, 3, 3), 1, 1, gw=gw)
    cv = Fn.cost_volume(f1, f2, lrelu=True)
    cv.backward(torch.ones_like(cv))
    wp = Fn.warp(f1, fl, 384, 448, 0.05)
    wp.backward(torch.ones_like(wp))
torch.cuda.synchronize()
