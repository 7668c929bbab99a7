"""Diagnostic: which h2 run is right?  Image gradient of h2 single-stream / h2 lane runs against an x3 single-stream and an fp32-MFMA
single-stream run of the same step."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_train_gpu import _setup, _batch
from irr_amd import conv as C

m, mal, arena, opt, step = _setup(4, lane=False)
b = _batch(4, 384, 448)


def grads(lane, math):
    C.set_math(math)
    if lane:
        arena.enable_async_wgrad()
    try:
        arena.zero_grad()
        for k in ("input1", "input2"):
            b[k].grad = None
            b[k].requires_grad_(True)
        ld, _ = mal(b)
        ld["total_loss"].backward()
        arena.sync()
        torch.cuda.synchronize()
        return torch.cat([b["input1"].grad.flatten(), b["input2"].grad.flatten()]).clone(), arena.flat.clone()
    finally:
        if lane:
            arena.disable_async_wgrad()
        C.set_math(C.DEFAULT_MATH)


def rel(a, b_):
    return ((a - b_).double().norm() / b_.double().norm()).item()


f32 = grads(False, "f32")
x3 = grads(False, "x3")
h2 = grads(False, "h2")
print(f"single stream: x3 vs f32 image {rel(x3[0], f32[0]):.2e} params {rel(x3[1], f32[1]):.2e}; h2 vs f32 image {rel(h2[0], f32[0]):.2e} params {rel(h2[1], f32[1]):.2e}; h2 vs x3 image {rel(h2[0], x3[0]):.2e}")
for it in range(10):
    g = grads(True, "h2")
    print(f"h2 lane run {it}: vs h2 single stream {rel(g[0], h2[0]):.2e}; vs x3 {rel(g[0], x3[0]):.2e}; vs f32 {rel(g[0], f32[0]):.2e} | params vs f32 {rel(g[1], f32[1]):.2e}", flush=True)
for it in range(4):
    g = grads(True, "x3")
    print(f"x3 lane run {it}: vs x3 single stream {rel(g[0], x3[0]):.2e}; vs f32 {rel(g[0], f32[0]):.2e}", flush=True)
