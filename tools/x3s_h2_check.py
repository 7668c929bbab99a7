"""The streaming 32-channel kernel in its two forms (bf16x3 / fp16x2): error against an fp64 convolution and time per launch for the
epilogue variants the OccUpsampleNetwork node uses (plain, residual, data gradient with mask / accumulate, dual output)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irr_amd import conv as C

C.set_x3s_h2(True)                                       # (off by default: conv.X3S_H2)

torch.manual_seed(0)
B, H, W = int(os.environ.get("B", 4)), 384, 448
dev = "cuda"


def rel(a, b):
    return ((a.double() - b).norm() / b.norm()).item()


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for scale in (1.0, 1e-5, 300.0):
    x = torch.randn(B, 32, H, W, device=dev) * scale
    x[:, :, :8] *= 1e-3                                      # a wide value range inside one tensor
    w = torch.randn(32, 32, 3, 3, device=dev) * 0.05
    bias = torch.randn(32, device=dev) * 0.1 * scale
    res = torch.randn(B, 32, H, W, device=dev) * scale
    mask = torch.randn(B, 32, H, W, device=dev)
    truth = torch.nn.functional.conv2d(x.double(), w.double(), bias.double(), padding=1)
    truth_lr = torch.nn.functional.leaky_relu(truth, 0.1)
    truth_d = torch.nn.functional.conv_transpose2d(x.double(), w.double(), padding=1)
    for math in ("x3", "h2"):
        C.set_math(math)
        xa = C.amax_measure(x) if math == "h2" else None
        ya = C.Amax.zeros(x.device, 1) if math == "h2" else None
        y = C.conv_forward(x, w, bias, 1, 1, True, x_amax=xa, y_amax=ya)
        line = f"scale {scale:g} {math}: fwd+lrelu {rel(y, truth_lr):.2e}"
        if ya is not None:
            torch.cuda.synchronize()
            line += f" (fused amax {'==' if ya.slots[ya.first].item() == y.abs().max().item() else '!='} max|y|)"
        y = C.conv_forward(x, w, bias, 1, 1, False, res=res, alpha=0.1, x_amax=xa)
        line += f"; res+0.1*conv {rel(y, res.double() + 0.1 * truth):.2e}"
        e, y2 = C.conv_forward_skip(x, w, bias, True, res, x_amax=xa)
        line += f"; dual e {rel(e, truth_lr):.2e} y {rel(y2, res.double() + truth_lr):.2e}"
        gx = res.clone()
        C.conv_dgrad(x, w, 1, 1, (H, W), gx=gx, accumulate=True, mask=mask, nmask=32, gy_amax=xa)
        td = (res.double() + truth_d) * torch.where(mask > 0, 1.0, 0.1).double()
        line += f"; dgrad acc+mask {rel(gx, td):.2e}"
        t0 = timeit(lambda: C.conv_forward(x, w, bias, 1, 1, True, x_amax=xa, y_amax=ya))
        t1 = timeit(lambda: C.conv_forward(x, w, bias, 1, 1, False, res=res, alpha=0.1, x_amax=xa))
        t2 = timeit(lambda: C.conv_dgrad(x, w, 1, 1, (H, W), gx=gx, accumulate=True, mask=mask, nmask=32, gy_amax=xa))
        t3 = timeit(lambda: C.conv_forward_skip(x, w, bias, True, res, x_amax=xa))
        print(line + f" | us: plain {t0:.0f} res {t1:.0f} dgrad acc+mask {t2:.0f} dual {t3:.0f}", flush=True)
# 16 -> 32 (init_conv on the zero-padded buffer) and 32 -> 64 / 64 -> 32 data gradients (two co-tile launches)
for cin, cout in ((16, 32), (32, 64)):
    x = torch.randn(B, cin, H, W, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    truth = torch.nn.functional.conv2d(x.double(), w.double(), None, padding=1)
    for math in ("x3", "h2"):
        C.set_math(math)
        code = C.x3_code(B, cin, H, W, cout, 3, 1, 1)
        xa = C.amax_measure(x) if math == "h2" else None
        y = C.conv_forward(x, w, None, 1, 1, False, x_amax=xa)
        t0 = timeit(lambda: C.conv_forward(x, w, None, 1, 1, False, x_amax=xa))
        print(f"{cin} -> {cout} {math} (code {code}): {rel(y, truth):.2e}, {t0:.0f} us", flush=True)
