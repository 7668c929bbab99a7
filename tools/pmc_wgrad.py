"""Workload for the round-6 PMC passes (VERDICT r5 missing #2 / next #3): the kernels that are on the DEFAULT path now -- the fp16x2
weight gradients (conv_wgrad_x3_kernel<..., false, true>) on the level-4 shapes and the 32-channel full-resolution layers, the
fp16x2 streaming kernel conv_x3s_kernel<0, 2> (plain forward writing bit masks) / <4, 2> (data gradient, accumulate + bit mask), and
the forward kernel on two shapes for reference.  Three launches each.  PMC_SET=wgrad|x3s|all"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irr_amd import conv as C
C.set_math("h2")
which = os.environ.get("PMC_SET", "all")
B, H, W = 64, 96, 112
if which in ("all", "wgrad"):
    for cin, cout in ((565, 128), (115, 128), (371, 96), (467, 64), (531, 32), (128, 64), (128, 128), (64, 64)):
        x = torch.randn(B, cin, H, W, device="cuda")
        gy = torch.randn(B, cout, H, W, device="cuda")
        gw = torch.zeros(cout, cin, 3, 3, device="cuda")
        xa, ga = C.amax_measure(x), C.amax_measure(gy)
        for _ in range(3):
            C.conv_wgrad(x, gy, (cout, cin, 3, 3), 1, 1, gw=gw, x_amax=xa, gy_amax=ga)
        if (cin, cout) in ((565, 128), (115, 128)):
            w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
            b = torch.randn(cout, device="cuda")
            for _ in range(3):
                C.conv_forward(x, w, b, 1, 1, True, x_amax=xa)
        del x, gy, gw
    torch.cuda.synchronize()
B2, H2, W2 = 16, 384, 448
x = torch.randn(B2, 32, H2, W2, device="cuda")
gy = torch.randn(B2, 32, H2, W2, device="cuda")
w = torch.randn(32, 32, 3, 3, device="cuda") * 0.05
b = torch.randn(32, device="cuda")
xa, ga = C.amax_measure(x), C.amax_measure(gy)
if which in ("all", "wgrad"):
    gw = torch.zeros(32, 32, 3, 3, device="cuda")
    for _ in range(3):
        C.conv_wgrad(x, gy, (32, 32, 3, 3), 1, 1, gw=gw, x_amax=xa, gy_amax=ga)
if which in ("all", "x3s"):
    assert C.x3s_bits_ok(B2, 32, H2, W2, 32)
    bits = torch.empty(C.x3s_mask_words(B2, H2, W2), dtype=torch.int32, device="cuda")
    gx = torch.randn(B2, 32, H2, W2, device="cuda")
    for _ in range(3):
        y = C.conv_forward(x, w, b, 1, 1, True, x_amax=xa, bits_out=bits)
        C.conv_dgrad(gy, w, 1, 1, (H2, W2), gx=gx, accumulate=True, mask=y, nmask=32, mask_bits=bits, gy_amax=ga)
torch.cuda.synchronize()
print(dict(C.LAUNCHES))
