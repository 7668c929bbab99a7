import os, sys, torch
sys.path.insert(0, "/root/repo")
sys.path.insert(0, os.getcwd())
from irr_amd import conv as C
from tools.x3_check import timeit
B = 64
gy = torch.randn(B, 16, 192, 224, device="cuda"); w = torch.randn(16, 3, 3, 3, device="cuda") * 0.1
for thr, name in ((1 << 30, "2x2-block kernel"), (0, "zero-interleave + MFMA conv")):
    C.S2_GATHER_MAX_CIN = thr
    t = timeit(lambda: C.conv_dgrad(gy, w, 2, 1, (384, 448)))
    print(f"first-conv image gradient (16 -> 3, stride 2, 384x448x64), {name}: {t:.3f} ms")
for thr, name in ((1 << 30, "2x2-block kernel"), (0, "zero-interleave + MFMA conv")):
  C.S2_GATHER_MAX_CIN = thr
  print(name)
  tot = 0.0
  for cout, cin, H, W in ((32, 16, 192, 224), (64, 32, 96, 112), (96, 64, 48, 56), (128, 96, 24, 28), (196, 128, 12, 14)):
    gy = torch.randn(B, cout, H // 2, W // 2, device="cuda"); w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.1
    t = timeit(lambda: C.conv_dgrad(gy, w, 2, 1, (H, W)))
    gf = 2.0 * B * (H // 2) * (W // 2) * cout * cin * 9 / 1e9
    tot += t
    print(f"  stride-2 data gradient {cout:3d} -> {cin:3d} to {H}x{W}: {t:.3f} ms  ({gf / t:.1f} TFLOP/s of useful work)")
  print(f"  total {tot:.3f} ms per step")
