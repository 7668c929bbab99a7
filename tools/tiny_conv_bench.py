"""Forward launches of the decoder / context / refinement layer shapes at the 6x7 and 12x14 pyramid levels (2B = 64 samples): time, TFLOP/s and
the kernel family the routing picks (code 0 = fp32 split-K kernel, 4171 ... = conv_x3_kernel + K-split epilogue).  profiles/NOTES.md,
section C: a dedicated tiny-plane bf16x3 kernel was measured against these numbers and removed."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irr_amd import conv as C
from tools.x3_check import timeit
SHAPES = [(565, 128, 1), (115, 128, 1), (243, 128, 1), (371, 96, 1), (467, 64, 1), (531, 32, 1), (128, 128, 1), (128, 128, 2), (128, 128, 4),
          (128, 96, 8), (96, 64, 16), (64, 32, 1), (128, 64, 1), (448, 115, 1), (196, 196, 1)]
tot = {}
for H, W in ((6, 7), (12, 14)):
    for cin, cout, dil in SHAPES:
        x = torch.randn(64, cin, H, W, device="cuda")
        w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
        b = torch.zeros(cout, device="cuda")
        code = C.x3_code(64, cin, H, W, cout, 3, 1, dil)
        t = timeit(lambda: C.conv_forward(x, w, b, 1, dil, True), iters=20)
        gf = 2.0 * 64 * H * W * cin * cout * 9 / 1e9
        tot[(H, W)] = tot.get((H, W), 0.0) + t
        print(f"{H:2d}x{W:2d} {cin:4d}->{cout:4d} d{dil:2d}: {t * 1e3:7.1f} us  {gf / t:6.1f} TFLOP/s  code {code}", flush=True)
print({k: f"{v * 1e3:.0f} us" for k, v in tot.items()})
