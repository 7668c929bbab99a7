"""Warp backward (gather for the flow gradient + atomic scatter for the image gradient) at the 96x112 level, 2B = 64 samples."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irr_amd import functional as Fn
from tools.x3_check import timeit
B, H, W = 64, 96, 112
for C in (32, 3):
    x = torch.randn(B, C, H, W, device="cuda", requires_grad=True)
    for amp in (0.02, 0.5):
        fl = (torch.randn(B, 2, H, W, device="cuda") * amp).requires_grad_(True)
        y = Fn.warp(x, fl, 384, 448, 0.05)
        go = torch.randn_like(y)
        tb = timeit(lambda: torch.autograd.grad(y, (x, fl), go, retain_graph=True), iters=10)
        tx = timeit(lambda: torch.autograd.grad(y, (x,), go, retain_graph=True), iters=10)
        tf = timeit(lambda: torch.autograd.grad(y, (fl,), go, retain_graph=True), iters=10)
        Fn._WARP_BWD_ATOMIC = True                      # round 3's route: device-scope atomic scatter
        ta = timeit(lambda: torch.autograd.grad(y, (x, fl), go, retain_graph=True), iters=10)
        tax = timeit(lambda: torch.autograd.grad(y, (x,), go, retain_graph=True), iters=10)
        Fn._WARP_BWD_ATOMIC = False
        print(f"warp backward C={C:2d} 96x112x64, flow noise {amp}: both {tb*1e3:7.1f} us, image gradient only {tx*1e3:7.1f} us, flow gradient "
              f"only {tf*1e3:7.1f} us | atomic scatter (IRR_WARP_BWD_ATOMIC=1): both {ta*1e3:7.1f} us, image gradient only {tax*1e3:7.1f} us")
