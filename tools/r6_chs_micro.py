"""weight-gradient launches with and without the per-channel scale of the gy-role operand (channel maxima given: no pass timed)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irr_amd import conv as C, hip
from tools.x3_check import timeit
C.set_math("h2")
for name, cin, cout, dil, B, H, W in [("L3 467->64", 467, 64, 1, 64, 48, 56), ("L3 243->128", 243, 128, 1, 64, 48, 56), ("L3 128->64", 128, 64, 1, 64, 48, 56),
                                      ("L2 467->64", 467, 64, 1, 64, 24, 28), ("L4 467->64", 467, 64, 1, 64, 96, 112), ("L4 565->128", 565, 128, 1, 64, 96, 112),
                                      ("L4 371->96", 371, 96, 1, 64, 96, 112), ("L4 531->32", 531, 32, 1, 64, 96, 112)]:
    x = torch.randn(B, cin, H, W, device="cuda"); gy = torch.randn(B, cout, H, W, device="cuda"); gw = torch.zeros(cout, cin, 3, 3, device="cuda")
    xa, ga = C.amax_measure(x), C.amax_measure(gy)
    xc, gc = C.channel_amax(x), C.channel_amax(gy)
    C.WGRAD_CHANNEL_SCALE = False
    t0 = timeit(lambda: C.conv_wgrad(x, gy, gw.shape, 1, dil, gw=gw, x_amax=xa, gy_amax=ga))
    C.WGRAD_CHANNEL_SCALE = True
    t1 = timeit(lambda: C.conv_wgrad(x, gy, gw.shape, 1, dil, gw=gw, x_amax=xa, gy_amax=ga, x_chmax=xc, gy_chmax=gc))
    tp = timeit(lambda: C.channel_amax(gy))
    side = "x robust" if hip.lib().irr_conv2d_wgrad_h2_robust_side(B, cin, H, W, cout, dil) else "gy robust (roles exchanged)"
    print(f"{name:14s} tensor scale {t0 * 1e3:7.1f} us | channel scales {t1 * 1e3:7.1f} us ({t1 / t0 - 1:+.1%}) | pass over gy {tp * 1e3:6.1f} us   [{side}]")
