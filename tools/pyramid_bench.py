"""Forward / weight-gradient / data-gradient time of the twelve feature-pyramid convs at the BASELINE batch (2B = 64 samples)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irr_amd import conv as C
from tools.x3_check import timeit
B, H, W = 64, 384, 448
chs = [3, 16, 32, 64, 96, 128, 196]
tot = [0.0, 0.0, 0.0]
for l in range(6):
    cin, cout = chs[l], chs[l + 1]
    for (ci, co, st, h, w) in ((cin, cout, 2, H >> l, W >> l), (cout, cout, 1, H >> (l + 1), W >> (l + 1))):
        x = torch.randn(B, ci, h, w, device="cuda"); wt = torch.randn(co, ci, 3, 3, device="cuda") * 0.05; b = torch.randn(co, device="cuda")
        oh, ow = C.out_hw(h, w, 3, st, 1)
        gy = torch.randn(B, co, oh, ow, device="cuda"); gw = torch.zeros_like(wt)
        tf = timeit(lambda: C.conv_forward(x, wt, b, st, 1, True))
        tw = timeit(lambda: C.conv_wgrad(x, gy, wt.shape, st, 1, gw=gw))
        td = timeit(lambda: C.conv_dgrad(gy, wt, st, 1, (h, w)))
        gf = 2.0 * B * oh * ow * co * ci * 9 / 1e9
        tot[0] += tf; tot[1] += tw; tot[2] += td
        print(f"{ci:3d} -> {co:3d} stride {st} at {h}x{w}: fwd {tf:6.3f} ms ({gf / tf:5.1f} TF)  wgrad {tw:6.3f} ms ({gf / tw:5.1f} TF)  dgrad {td:6.3f} ms ({gf / td:5.1f} TF)", flush=True)
print(f"total: fwd {tot[0]:.2f} ms, wgrad {tot[1]:.2f} ms, dgrad {tot[2]:.2f} ms")
