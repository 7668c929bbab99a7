"""Lane vs single-stream gradients at 4 x 384x448, repeated (round 4, profiles/NOTES.md section C): the gradient with respect to the
images -- main-stream work only -- and every parameter gradient of N two-stream passes against one single-stream pass.
    python tools/lane_race_probe.py [N]            IRR_LANE_MAX_LEAD=0: unbounded lead (the state in which deviations were found)
    IRR_CONV_MATH=x3 ...                           the bf16x3 form (deviating passes are rare there)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_train_gpu import _setup, _batch
from irr_amd import conv as C

m, mal, arena, opt, step = _setup(4, lane=False)
b = _batch(4, 384, 448)


def grads(lane):
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
        return arena.flat.clone(), torch.cat([b["input1"].grad.flatten(), b["input2"].grad.flatten()]).clone()
    finally:
        if lane:
            arena.disable_async_wgrad()


def rel(a, b_):
    return ((a - b_).double().norm() / (b_.double().norm() + 1e-300)).item()


ref, ref_img = grads(False)
print("math", C.MATH, "IRR_LANE_MAX_LEAD", os.environ.get("IRR_LANE_MAX_LEAD", "1 (default)"), "IRR_LANE_GROUP", os.environ.get("IRR_LANE_GROUP", "4 (default)"))
for it in range(4):
    g, img = grads(False)
    print(f"single stream {it}: image gradient {rel(img, ref_img):.2e}, parameters {rel(g, ref):.2e}", flush=True)
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    g, img = grads(True)
    off, worst = 0, (0.0, "")
    for n, p in m.named_parameters():
        k = p.numel()
        e = rel(g[off:off + k], ref[off:off + k])
        worst = max(worst, (e, n))
        off += k
    print(f"lane {it}: image gradient {rel(img, ref_img):.2e}, parameters {rel(g, ref):.2e}, worst parameter {worst[1]} {worst[0]:.1e}", flush=True)
