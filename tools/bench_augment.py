"""Times the two augmentation kernels on the FlyingChairsOcc training shape (B x 384 x 512) with HIP events and prints
algorithmic GB/s (images: 3 ch read + 3 ch written; flow+occ: 3 ch read + 3 ch written; fp32) next to the 8 TB/s HBM peak.
The oracle's CPU time for the same batch is printed beside it (bounded: B=2)."""
import argparse
import os
import sys
import time
import types

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from irr_amd import augment as A  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--height", type=int, default=384)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    B, H, W = a.batch, a.height, a.width
    dev = "cuda:0"
    torch.manual_seed(0)
    np.random.seed(0)
    ex = {"input1": torch.rand(B, 3, H, W, device=dev), "input2": torch.rand(B, 3, H, W, device=dev),
          "target1": 5 * torch.randn(B, 2, H, W, device=dev), "target2": 5 * torch.randn(B, 2, H, W, device=dev),
          "target_occ1": (torch.rand(B, 1, H, W, device=dev) < 0.3).float(),
          "target_occ2": (torch.rand(B, 1, H, W, device=dev) < 0.3).float()}
    aug = A.RandomAffineFlowOcc(types.SimpleNamespace(), addnoise=True)
    th1, th2 = aug.sample(B, H, W)
    inv1 = A.invert_thetas(th1).cuda()
    t1, t2 = th1.cuda(), th2.cuda()
    nz = torch.randn(B, 3, H, W, device=dev)

    def timed(fn):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.iters * 1e-3

    px = B * H * W
    tw = timed(lambda: A.affine_warp(ex["input1"], inv1, None, nz, 0.02))
    tf = timed(lambda: A.affine_flow_occ(ex["target1"], ex["target_occ1"], inv1, t1, t2, None))
    t0 = time.perf_counter()
    for _ in range(10):
        aug(dict(ex))
    torch.cuda.synchronize()
    tall = (time.perf_counter() - t0) / 10
    print(f"B={B} {H}x{W}")
    print(f"irr_affine_warp_f32 (+noise)  {tw*1e6:8.1f} us  {px*(3+3+3)*4/tw/1e9:8.1f} GB/s algorithmic  (peak 8000)")
    print(f"irr_affine_flow_occ_f32       {tf*1e6:8.1f} us  {px*6*4/tf/1e9:8.1f} GB/s algorithmic  (peak 8000)")
    print(f"RandomAffineFlowOcc.forward   {tall*1e3:8.2f} ms per batch incl. host sampling + randn  ({B/tall:.0f} pairs/s)")
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
    import augment_oracle as AO
    cpu = {k: v[:2].cpu() for k, v in ex.items()}
    t0 = time.perf_counter()
    AO.random_affine_flow_occ(cpu, addnoise=True)
    tc = time.perf_counter() - t0
    print(f"oracle (CPU torch, {torch.get_num_threads()} threads) B=2: {tc*1e3:.1f} ms  ({2/tc:.1f} pairs/s)")


if __name__ == "__main__":
    main()
