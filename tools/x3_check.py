"""Accuracy (vs an fp64 CPU convolution) and speed of the bf16x3-split conv kernel next to the fp32-MFMA kernel."""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irr_amd import conv as C, hip  # noqa: E402

ACC = [  # cin, cout, dil, B, H, W
    (115, 128, 1, 2, 24, 28), (565, 128, 1, 1, 16, 48), (371, 96, 1, 1, 33, 47), (531, 32, 1, 2, 20, 36),
    (128, 64, 1, 1, 40, 24), (32, 32, 1, 1, 70, 90), (128, 128, 2, 1, 24, 28), (128, 128, 4, 1, 24, 28), (16, 565, 1, 1, 24, 28),
    (243, 128, 1, 1, 48, 56),
    (128, 96, 8, 1, 48, 56), (96, 64, 16, 1, 48, 56), (96, 128, 8, 2, 27, 36), (64, 96, 16, 1, 50, 44), (128, 128, 2, 1, 25, 28),
    (128, 128, 4, 2, 30, 36), (96, 64, 16, 2, 96, 112),
]
PERF = [  # name, cin, cout, dil, B, H, W
    ("ctx.conv0 L4", 565, 128, 1, 64, 96, 112), ("dense.conv1 L4", 115, 128, 1, 64, 96, 112),
    ("dense.conv3 L4", 371, 96, 1, 64, 96, 112), ("dense.conv5 L4", 531, 32, 1, 64, 96, 112),
    ("refine 128->64 L4", 128, 64, 1, 64, 96, 112), ("dense.conv4 L4", 467, 64, 1, 64, 96, 112), ("refine 64->64 L4", 64, 64, 1, 64, 96, 112), ("dense.conv4 L3", 467, 64, 1, 64, 48, 56), ("ctx d2 L4", 128, 128, 2, 64, 96, 112), ("ctx d4 L4", 128, 128, 4, 64, 96, 112),
    ("occup 32->32 L6", 32, 32, 1, 32, 384, 448), ("occup 32->32 L5", 32, 32, 1, 64, 192, 224),
    ("dense.conv2 L3", 243, 128, 1, 64, 48, 56), ("dgrad ctx0 L4", 128, 565, 1, 64, 96, 112),
    ("dense.conv2 L2", 243, 128, 1, 64, 24, 28), ("ctx.conv0 L2", 565, 128, 1, 64, 24, 28), ("dense.conv4 L2", 467, 64, 1, 64, 24, 28), ("refine 32->64 L4", 32, 64, 1, 64, 96, 112),
    ("refine 64->32 dgrad L4", 64, 32, 1, 64, 96, 112),
    ("ctx d8 L4", 128, 96, 8, 64, 96, 112), ("ctx d16 L4", 96, 64, 16, 64, 96, 112), ("ctx d8 dgrad L4", 96, 128, 8, 64, 96, 112),
    ("ctx d16 dgrad L4", 64, 96, 16, 64, 96, 112), ("ctx d8 L3", 128, 96, 8, 64, 48, 56), ("ctx d2 L3", 128, 128, 2, 64, 48, 56),
    ("ctx d4 L3", 128, 128, 4, 64, 48, 56), ("ctx d16 L3", 96, 64, 16, 64, 48, 56),
    ("ksplit ctx.conv0 L1", 565, 128, 1, 64, 12, 14), ("ksplit dense.conv1 L1", 115, 128, 1, 64, 12, 14), ("ksplit dense.conv2 L1", 243, 128, 1, 64, 12, 14),
    ("ksplit dense.conv3 L1", 371, 96, 1, 64, 12, 14), ("ksplit dense.conv4 L1", 467, 64, 1, 64, 12, 14), ("ksplit ctx d2 L1", 128, 128, 2, 64, 12, 14),
    ("ksplit dgrad ctx0 L1", 128, 565, 1, 64, 12, 14), ("ksplit dense.conv4 L2b", 467, 64, 1, 64, 24, 28), ("ksplit dense.conv3 L2b", 371, 96, 1, 64, 24, 28),
]
if os.environ.get("X3_ONLY"):
    PERF = [p for p in PERF if os.environ["X3_ONLY"] in p[0]]


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    hip.lib().irr_conv_x3_set_min_blocks(0)
    print("== accuracy: max |err| / max |ref|, fp64 reference ==")
    for cin, cout, dil, B, H, W in ACC:
        g = torch.Generator().manual_seed(cin * 7 + cout)
        x = torch.randn(B, cin, H, W, generator=g) * torch.exp(torch.randn(B, cin, 1, 1, generator=g))
        w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
        b = torch.randn(cout, generator=g) * 0.1
        ref = F.conv2d(x.double(), w.double(), b.double(), padding=dil, dilation=dil)
        gy = torch.randn(B, cout, H, W, generator=g)
        gref = torch.nn.grad.conv2d_input(x.shape, w.double(), gy.double(), padding=dil, dilation=dil)
        out = {}
        for m in ("f32", "x3"):
            C.set_math(m)
            code = C.x3_code(B, cin, H, W, cout, 3, 1, dil)
            y = C.conv_forward(x.cuda(), w.cuda(), b.cuda(), 1, dil, False)
            gx = C.conv_dgrad(gy.cuda(), w.cuda(), 1, dil, (H, W))
            out[m] = ((y.cpu().double() - ref).abs().max().item() / ref.abs().max().item(),
                      (gx.cpu().double() - gref).abs().max().item() / gref.abs().max().item(), code)
        print(f"{cin:4d}->{cout:4d} d{dil} {B}x{H}x{W}: fwd f32 {out['f32'][0]:.2e} x3 {out['x3'][0]:.2e} | dgrad f32 {out['f32'][1]:.2e} "
              f"x3 {out['x3'][1]:.2e}  code {out['x3'][2]}", flush=True)
    hip.lib().irr_conv_x3_set_min_blocks(384)
    if "--noperf" in sys.argv:
        return
    print("== speed ==")
    for name, cin, cout, dil, B, H, W in PERF:
        x = torch.randn(B, cin, H, W, device="cuda")
        w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
        b = torch.randn(cout, device="cuda")
        gf = 2.0 * B * H * W * cout * cin * 9 / 1e9
        row = f"{name:20s} {gf:8.1f} GF "
        for m in ("f32", "x3"):
            C.set_math(m)
            t = timeit(lambda: C.conv_forward(x, w, b, 1, dil, True))
            row += f" {m}: {t:6.2f} ms {gf / t:6.1f} TF"
        row += f"  code {C.x3_code(B, cin, H, W, cout, 3, 1, dil)}"
        print(row, flush=True)


if __name__ == "__main__":
    main()
