"""Cost-volume forward/backward timing at the pyramid levels of BASELINE configs[2] (2B = 64 samples)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irr_amd import functional as Fn
from tools.x3_check import timeit
for C, H, W in [(32, 96, 112), (32, 48, 56), (32, 24, 28)]:
    B = 64
    f1 = torch.randn(B, C, H, W, device="cuda", requires_grad=True); f2 = torch.randn(B, C, H, W, device="cuda", requires_grad=True)
    with torch.no_grad():
        t = timeit(lambda: Fn.cost_volume(f1, f2, lrelu=True), iters=10)
    cv = Fn.cost_volume(f1, f2, lrelu=True); go = torch.randn_like(cv)
    tb = timeit(lambda: torch.autograd.grad(cv, (f1, f2), go, retain_graph=True), iters=10)
    # round 4: the consumers hand back the pre-activation gradient, the gradient kernels do not read the 81-plane output
    cvp = Fn.cost_volume(f1, f2, lrelu=True, grad_is_preactivation=True)
    tp = timeit(lambda: torch.autograd.grad(cvp, (f1, f2), go, retain_graph=True), iters=10)
    by = B * H * W * 4 * (2 * C + 81) / 1e9
    byb = B * H * W * 4 * (2 * C + 81 + 2 * C) / 1e9          # gout + f1 + f2 read, g1 + g2 written (both gradients)
    print(f"corr C={C} {H}x{W}: fwd {t * 1e3:7.1f} us {by / t:5.2f} TB/s | bwd (both gradients) mask in the kernel {tb * 1e3:7.1f} us "
          f"{byb / tb:5.2f} TB/s algorithmic | pre-masked gradient {tp * 1e3:7.1f} us {byb / tp:5.2f} TB/s algorithmic", flush=True)
