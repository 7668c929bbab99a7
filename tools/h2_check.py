"""Accuracy (vs fp64 CPU references) and speed of the fp16x2 ("h2") form of conv_x3_kernel / conv_wgrad_x3_kernel next to the
bf16x3 form and the fp32-MFMA kernels: forward, data gradient, weight gradient; operand ranges from benign to hostile."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irr_amd import conv as C, hip  # noqa: E402
from tools.x3_check import timeit  # noqa: E402

ACC = [  # cin, cout, dil, B, H, W
    (115, 128, 1, 2, 24, 28), (565, 128, 1, 1, 16, 48), (371, 96, 1, 1, 33, 47), (531, 32, 1, 2, 20, 36),
    (128, 64, 1, 1, 40, 24), (128, 128, 2, 1, 24, 28), (128, 128, 4, 2, 30, 36), (96, 64, 16, 2, 96, 112), (128, 96, 8, 1, 48, 56),
    (64, 64, 1, 2, 24, 28),
]
RANGES = {   # name -> (activation generator, gradient scale)
    "unit": (lambda g, s: torch.randn(s, generator=g), 1.0),
    "per-channel 1e-2..1e2": (lambda g, s: torch.randn(s, generator=g) * torch.exp(2.0 * torch.randn(s[0], s[1], 1, 1, generator=g)), 1.0),
    "tiny gradients (1e-7)": (lambda g, s: torch.randn(s, generator=g), 1e-7),
    "outlier 1e4 in x": (None, 1.0),
    "relu-sparse": (lambda g, s: torch.relu(torch.randn(s, generator=g)) * 3.0, 1e-3),
}
PERF = [("ctx.conv0 L4", 565, 128, 1, 64, 96, 112), ("dense.conv1 L4", 115, 128, 1, 64, 96, 112), ("dense.conv3 L4", 371, 96, 1, 64, 96, 112),
        ("dense.conv5 L4", 531, 32, 1, 64, 96, 112), ("dense.conv4 L4", 467, 64, 1, 64, 96, 112), ("refine 128->64 L4", 128, 64, 1, 64, 96, 112),
        ("ctx d2 L4", 128, 128, 2, 64, 96, 112), ("ctx d8 L4", 128, 96, 8, 64, 96, 112), ("ctx d16 L4", 96, 64, 16, 64, 96, 112),
        ("dgrad ctx0 L4", 128, 565, 1, 64, 96, 112), ("dense.conv2 L3", 243, 128, 1, 64, 48, 56), ("dense.conv2 L2", 243, 128, 1, 64, 24, 28),
        ("ksplit ctx.conv0 L1", 565, 128, 1, 64, 12, 14)]


def rel(a, ref):
    return ((a.cpu().double() - ref).abs().max() / ref.abs().max()).item()


def main():
    hip.lib().irr_conv_x3_set_min_blocks(0)
    print("== accuracy: max |err| / max |ref| against fp64 (forward | data gradient | weight gradient) ==")
    for rname, (gen, gscale) in ({} if "--noacc" in sys.argv else RANGES).items():
        print(f"-- operand range: {rname}")
        for cin, cout, dil, B, H, W in ACC:
            g = torch.Generator().manual_seed(cin * 7 + cout)
            if gen is None:
                x = torch.randn(B, cin, H, W, generator=g) * 1e-2
                x[0, 0, 3, 3] = 1e4
            else:
                x = gen(g, (B, cin, H, W))
            w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
            b = torch.randn(cout, generator=g) * 0.1
            gy = torch.randn(B, cout, H, W, generator=g) * gscale
            ref = F.conv2d(x.double(), w.double(), b.double(), padding=dil, dilation=dil)
            gref = torch.nn.grad.conv2d_input(x.shape, w.double(), gy.double(), padding=dil, dilation=dil)
            wref = torch.nn.grad.conv2d_weight(x.double(), w.shape, gy.double(), padding=dil, dilation=dil)
            row = f"{cin:4d}->{cout:4d} d{dil:<2d} {B}x{H}x{W}:"
            for m in ("f32", "x3", "h2"):
                C.set_math(m)
                xc, wc, gc = x.cuda(), w.cuda(), gy.cuda()
                y = C.conv_forward(xc, wc, b.cuda(), 1, dil, False)
                gx = C.conv_dgrad(gc, wc, 1, dil, (H, W))
                xa, ga = (C.amax_measure(xc), C.amax_measure(gc)) if m == "h2" else (None, None)
                gw = C.conv_wgrad(xc, gc, w.shape, 1, dil, x_amax=xa, gy_amax=ga)
                row += f"  {m} {rel(y, ref):.1e} {rel(gx, gref):.1e} {rel(gw, wref):.1e}"
            routed = dict(C.LAUNCHES)
            print(row + f"   [{'h2' if routed.get('fwd_h2') else '--'} {'h2' if routed.get('dgrad_h2') else '--'} {'h2' if routed.get('wgrad_h2') else '--'}]", flush=True)
            C.LAUNCHES.clear()
    hip.lib().irr_conv_x3_set_min_blocks(384)
    if "--noperf" in sys.argv:
        return
    print("== speed (forward with a known amax slot | weight gradient), TFLOP/s of algorithmic fp32 work ==")
    for name, cin, cout, dil, B, H, W in PERF:
        x = torch.randn(B, cin, H, W, device="cuda")
        w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
        b = torch.randn(cout, device="cuda")
        gy = torch.randn(B, cout, H, W, device="cuda")
        gw = torch.zeros_like(w)
        gf = 2.0 * B * H * W * cout * cin * 9 / 1e9
        row = f"{name:20s} {gf:8.1f} GF "
        for m in ("x3", "h2"):
            C.set_math(m)
            xa, ga = (C.amax_measure(x), C.amax_measure(gy)) if m == "h2" else (None, None)
            ya = C.Amax.zeros(x.device) if m == "h2" else None
            t = timeit(lambda: C.conv_forward(x, w, b, 1, dil, True, x_amax=xa, y_amax=ya))
            tw = timeit(lambda: C.conv_wgrad(x, gy, w.shape, 1, dil, gw=gw, x_amax=xa, gy_amax=ga))
            row += f" {m}: fwd {t:6.2f} ms {gf / t:6.1f} | wgrad {tw:6.2f} ms {gf / tw:6.1f}  "
        C.set_math("h2")
        t = timeit(lambda: C.amax_measure(x))
        row += f" amax(x) {t * 1e3:6.1f} us = {x.numel() * 4 / t / 1e9:5.2f} TB/s"
        print(row, flush=True)


if __name__ == "__main__":
    main()
