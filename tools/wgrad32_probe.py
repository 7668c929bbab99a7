"""Is the 32 -> 32 weight gradient at 384x448 waiting for HBM?  Time per sample for B = 2 ... 64 (operands of 44 MB ... 1.4 GB each: the
small ones stay resident in the 256 MB memory-side cache across the timed repetitions)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irr_amd import conv as C
from tools.x3_check import timeit
for B in (2, 4, 8, 16, 32, 64):
    x = torch.randn(B, 32, 384, 448, device="cuda"); gy = torch.randn(B, 32, 384, 448, device="cuda")
    gw = torch.zeros(32, 32, 3, 3, device="cuda"); gb = torch.zeros(32, device="cuda")
    t = timeit(lambda: C.conv_wgrad(x, gy, (32, 32, 3, 3), 1, 1, gw=gw, gbias=gb), iters=10)
    gf = 2.0 * B * 384 * 448 * 32 * 32 * 9 / 1e9
    print(f"B={B:2d}: {t * 1e3:7.1f} us  {t * 1e3 / B:6.2f} us/sample  {gf / t:6.1f} TFLOP/s  operands {2 * B * 32 * 384 * 448 * 4 / 1e6:6.0f} MB", flush=True)
