import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irr_amd import conv as C
from tools.test_x3 import timeit
B, cin, H, W = 64, 563, 96, 112
x = torch.randn(B, cin, H, W, device="cuda"); w = torch.randn(2, cin, 3, 3, device="cuda") * 0.05; b = torch.randn(2, device="cuda")
gy = torch.randn(B, 2, H, W, device="cuda"); gw = torch.zeros(2, cin, 3, 3, device="cuda"); gb = torch.zeros(2, device="cuda")
t1 = timeit(lambda: C.conv_forward(x, w, b, 1, 1, False))
t2 = timeit(lambda: C.conv_wgrad(x, gy, (2, cin, 3, 3), 1, 1, gw=gw, gbias=gb))
gbytes = B * cin * H * W * 4 / 1e9
print(f"smallco fwd {t1:.3f} ms ({gbytes / t1:.2f} TB/s)  wgrad {t2:.3f} ms ({gbytes / t2:.2f} TB/s)")
