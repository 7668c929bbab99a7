"""The 96 -> 64 dilation-16 layer of the context networks at the 48x56 level (2B = 64): forward, data gradient, weight gradient --
time and the kernel family the routing picks.  Switches: IRR_X3_MIN_EFF=65 (forward / data gradient on conv_x3_kernel with its 4 x 64
tile), IRR_WX3_NO_D16_KG1=1 (weight gradient back on the fp32 kernel)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irr_amd import conv as C, hip
from tools.x3_check import timeit
B, cin, cout, H, W, d = 64, 96, 64, 48, 56, 16
x = torch.randn(B, cin, H, W, device="cuda"); w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05; b = torch.zeros(cout, device="cuda")
gy = torch.randn(B, cout, H, W, device="cuda"); gw = torch.zeros_like(w); gb = torch.zeros(cout, device="cuda")
gf = 2.0 * B * H * W * cin * cout * 9 / 1e9
tf = timeit(lambda: C.conv_forward(x, w, b, 1, d, True), iters=20)
td = timeit(lambda: C.conv_dgrad(gy, w, 1, d, (H, W)), iters=20)
tw = timeit(lambda: C.conv_wgrad(x, gy, w.shape, 1, d, gw=gw, gbias=gb), iters=20)
print(f"96->64 d16 48x56x64: fwd {tf*1e3:6.1f} us {gf/tf:6.1f} TF (code {C.x3_code(B, cin, H, W, cout, 3, 1, d)}) | dgrad {td*1e3:6.1f} us {gf/td:6.1f} TF "
      f"(code {C.x3_code(B, cout, H, W, cin, 3, 1, d)}) | wgrad {tw*1e3:6.1f} us {gf/tw:6.1f} TF (x3 {hip.lib().irr_conv2d_wgrad_x3_eligible(B, cin, H, W, cout, 3, 1, d)})"
      f"   MIN_EFF={os.environ.get('IRR_X3_MIN_EFF')} NO_D16_KG1={os.environ.get('IRR_WX3_NO_D16_KG1')}")
