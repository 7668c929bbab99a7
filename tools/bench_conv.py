"""Micro-benchmark of the conv family on the heavy IRR-PWC layer shapes, HIP vs MIOpen.
usage: python tools/bench_conv.py [--batch 64] [--backend hip|miopen|both]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irr_amd import conv as C  # noqa: E402
from tools import torch_conv_backend as TB  # noqa: E402
import contextlib  # noqa: E402

SHAPES = [  # name, Cin, Cout, k, stride, dil, H, W
    ("dense.conv1 L4", 115, 128, 3, 1, 1, 96, 112),
    ("dense.conv3 L4", 371, 96, 3, 1, 1, 96, 112),
    ("dense.conv5 L4", 531, 32, 3, 1, 1, 96, 112),
    ("ctx.conv0 L4", 565, 128, 3, 1, 1, 96, 112),
    ("ctx.conv2 d4 L4", 128, 128, 3, 1, 4, 96, 112),
    ("ctx.conv4 d16 L4", 96, 64, 3, 1, 16, 96, 112),
    ("refine 128->64 L4", 128, 64, 3, 1, 1, 96, 112),
    ("occup 32->32 L6", 32, 32, 3, 1, 1, 384, 448),
    ("dense.conv2 L2", 243, 128, 3, 1, 1, 24, 28),
    ("dense.conv2 L0", 243, 128, 3, 1, 1, 6, 7),
    ("pyr 3->16 s2 L5", 3, 16, 3, 2, 1, 384, 448),
]


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
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--backend", default="both")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    backends = ["hip", "miopen"] if a.backend == "both" else [a.backend]
    torch.backends.cudnn.benchmark = True
    print(f"{'layer':22s} {'GFLOP':>8s} " + " ".join(f"{b+'.'+p:>14s}" for b in backends for p in ("fwd", "dgrad", "wgrad")))
    for name, cin, cout, k, st, dil, H, W in SHAPES:
        if a.only and a.only not in name:
            continue
        B = a.batch if H * W < 100000 else max(1, a.batch // 2)
        x = torch.randn(B, cin, H, W, device="cuda")
        w = torch.randn(cout, cin, k, k, device="cuda") * 0.05
        bias = torch.randn(cout, device="cuda")
        oh, ow = C.out_hw(H, W, k, st, dil)
        gy = torch.randn(B, cout, oh, ow, device="cuda")
        gf = 2.0 * B * oh * ow * cout * cin * k * k / 1e9
        row = f"{name:22s} {gf:8.1f} "
        for be in backends:
            with (TB.torch_convs() if be == "miopen" else contextlib.nullcontext()):
                t_f = timeit(lambda: C.conv_forward(x, w, bias, st, dil, True))
                t_d = timeit(lambda: C.conv_dgrad(gy, w, st, dil, (H, W)))
                gw = torch.zeros_like(w)
                t_w = timeit(lambda: C.conv_wgrad(x, gy, w.shape, st, dil, gw))
            row += " ".join(f"{t:6.2f}ms{gf / t:5.1f}TF" for t in (t_f, t_d, t_w)) + " "
        print(row, flush=True)


if __name__ == "__main__":
    main()
