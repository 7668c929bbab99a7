"""Speed of the Cout <= 2 head kernels (conv_last of the dense estimators) at the level-4 / level-3 shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irr_amd import conv as C
from tools.x3_check import timeit
for (B, cin, H, W, co) in ((64, 563, 96, 112, 2), (64, 562, 96, 112, 1), (64, 563, 48, 56, 2), (64, 563, 24, 28, 2), (64, 32, 96, 112, 2)):
    x = torch.randn(B, cin, H, W, device="cuda"); w = torch.randn(co, cin, 3, 3, device="cuda") * 0.05; b = torch.randn(co, device="cuda")
    gy = torch.randn(B, co, H, W, device="cuda"); gw = torch.zeros(co, cin, 3, 3, device="cuda"); gb = torch.zeros(co, device="cuda")
    gx = torch.zeros_like(x)
    t1 = timeit(lambda: C.conv_forward(x, w, b, 1, 1, False))
    t2 = timeit(lambda: C.conv_wgrad(x, gy, (co, cin, 3, 3), 1, 1, gw=gw, gbias=gb))
    t3 = timeit(lambda: C.conv_dgrad(gy, w, 1, 1, (H, W), gx=gx, accumulate=True, mask=x, nmask=32))
    gbytes = B * cin * H * W * 4 / 1e9
    print(f"{cin}->{co} {B}x{H}x{W}: fwd {t1:.3f} ms ({gbytes / t1:.2f} TB/s)  wgrad {t2:.3f} ms ({gbytes / t2:.2f} TB/s)  "
          f"dgrad+= {t3:.3f} ms ({2 * gbytes / t3:.2f} TB/s rd+wr)", flush=True)
