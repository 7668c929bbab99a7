import os, sys, torch
sys.path.insert(0, "/root/repo")
sys.path.insert(0, os.getcwd())
from irr_amd import conv as C
from tools.x3_check import timeit
B = 64
gy = torch.randn(B, 16, 192, 224, device="cuda"); w = torch.randn(16, 3, 3, 3, device="cuda") * 0.1
for thr, name in ((4, "gather kernel"), (0, "zero-interleave + MFMA conv")):
    C.S2_GATHER_MAX_CIN = thr
    t = timeit(lambda: C.conv_dgrad(gy, w, 2, 1, (384, 448)))
    print(f"first-conv image gradient (16 -> 3, stride 2, 384x448x64), {name}: {t:.3f} ms")
