"""Accuracy (vs fp64) and speed of the bf16x3-split wgrad kernel next to the fp32-MFMA wgrad kernels."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irr_amd import conv as C, hip  # noqa: E402
from tools.x3_check import timeit  # noqa: E402

ACC = [  # cin, cout, B, H, W
    (115, 128, 2, 24, 32), (40, 128, 1, 16, 48), (371, 96, 1, 12, 56), (64, 128, 2, 8, 40), (35, 96, 1, 20, 64), (565, 128, 1, 8, 112),
    (128, 128, 2, 16, 16), (467, 64, 1, 16, 48), (64, 64, 2, 12, 56), (531, 32, 1, 16, 48), (64, 9, 2, 12, 56), (300, 32, 1, 8, 64), (32, 32, 2, 24, 64), (32, 9, 1, 16, 96), (24, 32, 2, 20, 32), (243, 128, 2, 24, 28), (11, 32, 2, 16, 64), (64, 64, 1, 12, 44), (371, 96, 1, 10, 36), (531, 32, 1, 24, 28),
]
ACC += [  # round 3: widths that are not a multiple of four, heights below 8 (tall-image walk), odd batches
    (565, 128, 8, 6, 7), (128, 128, 4, 12, 14), (371, 96, 6, 6, 7), (64, 32, 8, 12, 14), (32, 32, 8, 6, 7), (115, 128, 5, 12, 14),
    (64, 64, 3, 6, 7), (467, 64, 7, 12, 14), (531, 32, 9, 6, 7), (40, 128, 6, 5, 10), (64, 96, 5, 3, 9), (35, 128, 11, 4, 20), (128, 64, 3, 9, 22),
]
ACC_DIL = [(128, 128, 1, 24, 32, 2), (128, 128, 2, 19, 48, 4), (128, 96, 1, 40, 56, 8), (96, 64, 1, 50, 48, 16), (128, 128, 1, 20, 56, 2),
           (128, 128, 1, 33, 64, 4), (128, 96, 2, 24, 64, 8), (96, 64, 2, 40, 64, 16), (96, 64, 2, 48, 56, 16), (96, 64, 1, 35, 40, 16)]
PERF_DIL = [("ctx d2 L4", 128, 128, 64, 96, 112, 2), ("ctx d4 L4", 128, 128, 64, 96, 112, 4), ("ctx d8 L4", 128, 96, 64, 96, 112, 8),
            ("ctx d16 L4", 96, 64, 64, 96, 112, 16), ("ctx d2 L3", 128, 128, 64, 48, 56, 2), ("ctx d8 L3", 128, 96, 64, 48, 56, 8), ("ctx d16 L3", 96, 64, 64, 48, 56, 16),
            ("ctx d2 L2", 128, 128, 64, 24, 28, 2), ("ctx d4 L2", 128, 128, 64, 24, 28, 4), ("ctx d8 L2", 128, 96, 64, 24, 28, 8), ("ctx d16 L2", 96, 64, 64, 24, 28, 16)]
PERF_SMALL = [("ctx.conv0 L1", 565, 128, 64, 12, 14), ("refine 128->128 L1", 128, 128, 64, 12, 14), ("dense.conv4 L1", 467, 64, 64, 12, 14),
              ("dense.conv5 L1", 531, 32, 64, 12, 14), ("refine 64->32 L1", 64, 32, 64, 12, 14), ("refine 32->32 L1", 32, 32, 64, 12, 14),
              ("ctx.conv0 L0", 565, 128, 64, 6, 7), ("refine 128->128 L0", 128, 128, 64, 6, 7), ("dense.conv4 L0", 467, 64, 64, 6, 7),
              ("dense.conv5 L0", 531, 32, 64, 6, 7), ("refine 64->32 L0", 64, 32, 64, 6, 7), ("refine 32->32 L0", 32, 32, 64, 6, 7)]
PERF = [("ctx.conv0 L4", 565, 128, 64, 96, 112), ("dense.conv1 L4", 115, 128, 64, 96, 112), ("dense.conv3 L4", 371, 96, 64, 96, 112),
        ("refine 128->128 L4", 128, 128, 64, 96, 112), ("dense.conv2 L3", 243, 128, 64, 48, 56), ("ctx.conv0 L3", 565, 128, 64, 48, 56),
        ("128->128 448x1024 L4", 128, 128, 16, 112, 256), ("dense.conv4 L4", 467, 64, 64, 96, 112), ("refine 128->64 L4", 128, 64, 64, 96, 112),
        ("refine 64->64 L4", 64, 64, 64, 96, 112), ("dense.conv5 L4", 531, 32, 64, 96, 112), ("refine 64->32 L4", 64, 32, 64, 96, 112),
        ("dense.conv5 L3", 531, 32, 64, 48, 56), ("refine 32->32 L4", 32, 32, 64, 96, 112), ("refine 32->32 L3", 32, 32, 64, 48, 56), ("refine 64->32 L3", 64, 32, 64, 48, 56), ("dgradlike 32->64 L4", 32, 64, 64, 96, 112), ("dense.conv3 L3", 371, 96, 64, 48, 56), ("dense.conv4 L3", 467, 64, 64, 48, 56), ("dense.conv1 L3", 115, 128, 64, 48, 56), ("occup 32->32 L6", 32, 32, 64, 384, 448), ("occup 32->32 L5", 32, 32, 64, 192, 224), ("occup init 11->32 L6", 11, 32, 64, 384, 448), ("dense.conv2 L2", 243, 128, 64, 24, 28), ("ctx.conv0 L2", 565, 128, 64, 24, 28), ("dense.conv4 L2", 467, 64, 64, 24, 28)]


def main():
    hip.lib().irr_conv_x3_set_min_blocks(0)
    print("== accuracy: max |err| / max |ref| (fp64 reference) ==")
    for case in ACC + ACC_DIL:
        cin, cout, B, H, W = case[:5]
        dil = case[5] if len(case) > 5 else 1
        g = torch.Generator().manual_seed(cin + cout)
        x = torch.randn(B, cin, H, W, generator=g)
        gy = torch.randn(B, cout, H, W, generator=g)
        ref = torch.nn.grad.conv2d_weight(x.double(), (cout, cin, 3, 3), gy.double(), padding=dil, dilation=dil)
        bref = gy.double().sum(dim=(0, 2, 3))
        out = {}
        for m in ("f32", "x3"):
            C.set_math(m)
            gw = torch.zeros(cout, cin, 3, 3, device="cuda")
            gb = torch.zeros(cout, device="cuda")
            C.conv_wgrad(x.cuda(), gy.cuda(), (cout, cin, 3, 3), 1, dil, gw=gw, gbias=gb, alpha=0.5)
            out[m] = ((2 * gw.cpu().double() - ref).abs().max().item() / ref.abs().max().item(),
                      (2 * gb.cpu().double() - bref).abs().max().item() / bref.abs().max().item())
        el = hip.lib().irr_conv2d_wgrad_x3_eligible(B, cin, H, W, cout, 3, 1, dil)
        print(f"{cin:4d}->{cout:4d} d{dil} {B}x{H}x{W}: gw f32 {out['f32'][0]:.2e} x3 {out['x3'][0]:.2e} | gb f32 {out['f32'][1]:.2e} x3 {out['x3'][1]:.2e}  code {el}", flush=True)
    hip.lib().irr_conv_x3_set_min_blocks(384)
    if "--noperf" in sys.argv:
        return
    print("== speed ==")
    for case in (PERF_SMALL if "--small" in sys.argv else PERF_DIL + PERF + PERF_SMALL):
        if os.environ.get("WX3_ONLY") and os.environ["WX3_ONLY"] not in case[0]:
            continue
        name, cin, cout, B, H, W = case[:6]
        dil = case[6] if len(case) > 6 else 1
        x = torch.randn(B, cin, H, W, device="cuda")
        gy = torch.randn(B, cout, H, W, device="cuda")
        gw = torch.zeros(cout, cin, 3, 3, device="cuda")
        gb = torch.zeros(cout, device="cuda")
        gf = 2.0 * B * H * W * cout * cin * 9 / 1e9
        row = f"{name:22s} {gf:8.1f} GF "
        for m in ("f32", "x3"):
            C.set_math(m)
            t = timeit(lambda: C.conv_wgrad(x, gy, (cout, cin, 3, 3), 1, dil, gw=gw, gbias=gb))
            row += f" {m}: {t:6.2f} ms {gf / t:6.1f} TF"
        print(row + f"  code {hip.lib().irr_conv2d_wgrad_x3_eligible(B, cin, H, W, cout, 3, 1, dil)}", flush=True)


if __name__ == "__main__":
    main()
