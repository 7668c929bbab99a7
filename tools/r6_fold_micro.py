"""cost of the channel-maxima fold in conv_x3_kernel's epilogue, per launch: conv_forward / conv_dgrad with and without y_chmax / gx_chmax
on the layer shapes that fold in the train step (bs32 384x448: 2B = 64 samples)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from irr_amd import conv as C

def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

CASES = [("dense.conv4", 467, 64, 1), ("dense.conv3", 371, 96, 1), ("dense.conv2", 243, 128, 1), ("ctx.conv4 d16", 96, 64, 16), ("ctx.conv3 d8", 128, 96, 8),
         ("ctx.conv1 d2", 128, 128, 2), ("column K=96", 96, 96, 1), ("column K=192", 192, 128, 1)]
for (H, W) in ((96, 112), (48, 56), (24, 28)):
    for name, cin, cout, dil in CASES:
        x = torch.randn(64, cin, H, W, device="cuda")
        w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
        code = C.h2_code(64, cin, H, W, cout, 3, 1, dil)
        if not code:
            continue
        xa = C.amax_measure(x)
        y = torch.empty(64, cout, H, W, device="cuda")
        ya = C.Amax.zeros(x.device, 1)
        ch = C.zero_slots(x.device, cout)
        t0 = t(lambda: C.conv_forward(x, w, None, 1, dil, True, out=y, x_amax=xa, y_amax=ya))
        t1 = t(lambda: C.conv_forward(x, w, None, 1, dil, True, out=y, x_amax=xa, y_amax=ya, y_chmax=ch))
        print(f"{H:3d}x{W:<3d} {name:14s} {cin:3d}->{cout:3d} d{dil:<2d} code {code:5d}  plain {t0:8.1f} us   + channel fold {t1:8.1f} us   {100 * (t1 / t0 - 1):+5.1f} %")
