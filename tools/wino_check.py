"""Round 6, VERDICT r5 next #1 -- the GATE of the Winograd F(2x2, 3x3) experiment on the fp16x2 arithmetic (csrc/conv_wino.hip):
(a) error against an fp64 convolution next to the direct fp16x2 kernel and the fp32-MFMA kernel, in the operand ranges and the
regional cases of tests/test_h2_gpu.py (bar: <= 4x the fp32-MFMA kernel, floor 1e-6); (b) time per launch at 96x112x64 for
565 -> 128 and 243 -> 128 (and the narrower layers) against conv_x3_kernel on the same operands (bar: >= 1.5x faster).
    python tools/wino_check.py [--noacc] [--noperf]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irr_amd import conv as C, hip  # noqa: E402


def pack(w, transpose=False):
    cout, cin = (w.shape[1], w.shape[0]) if transpose else (w.shape[0], w.shape[1])
    nbytes = int(hip.lib().irr_conv_wino_packed_bytes(cin, cout))
    uq = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    wmax = w.abs().max().reshape(1).float()
    hip.call("irr_conv_pack_weights_wino_h2", hip.ptr(w), uq.data_ptr(), cin, cout, int(transpose), hip.ptr(wmax), hip.stream())
    return uq, cin, cout


def wino_forward(x, packed, bias, lrelu, x_amax, y_amax=None, out=None):
    uq, cin, cout = packed
    B, _, H, W = x.shape
    y = out if out is not None else torch.empty(B, cout, H, W, device=x.device, dtype=torch.float32)
    hip.call("irr_conv2d_wino_fwd_h2", hip.ptr(x), uq.data_ptr(), hip.ptr(bias), hip.ptr(y), B, cin, H, W, cout, hip.bs(x), hip.bs(y),
             int(lrelu), 1.0, x_amax.ptr(), x_amax.n, y_amax.ptr() if y_amax is not None else None, hip.stream())
    return y


def timeit(fn, n=8):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def rel(a, ref):
    return ((a.cpu().double() - ref).abs().max() / ref.abs().max()).item()


RANGES = {
    "unit": lambda g, s: torch.randn(s, generator=g),
    "per-channel 1e-2..1e2": lambda g, s: torch.randn(s, generator=g) * torch.exp(2.0 * torch.randn(s[0], s[1], 1, 1, generator=g)),
    "tiny (1e-7)": lambda g, s: torch.randn(s, generator=g) * 1e-7,
    "outlier 1e4": None,
    "relu-sparse": lambda g, s: torch.relu(torch.randn(s, generator=g)) * 3.0,
    "3e12": lambda g, s: torch.randn(s, generator=g) * 3e12,
}
ACC = [(115, 128, 2, 24, 28), (565, 128, 1, 16, 48), (243, 128, 2, 24, 28), (128, 64, 1, 40, 24), (371, 96, 1, 33, 48), (64, 32, 2, 17, 20)]
PERF = [("ctx.conv0 L4", 565, 128, 64, 96, 112), ("dense.conv2 L4", 243, 128, 64, 96, 112), ("dense.conv1 L4", 115, 128, 64, 96, 112),
        ("dense.conv3 L4", 371, 96, 64, 96, 112), ("dense.conv4 L4", 467, 64, 64, 96, 112), ("dense.conv5 L4", 531, 32, 64, 96, 112),
        ("refine 128->64 L4", 128, 64, 64, 96, 112), ("dgrad-shaped 128->565 L4", 128, 565, 64, 96, 112),
        ("dense.conv2 L3", 243, 128, 64, 48, 56), ("ctx.conv0 L3", 565, 128, 64, 48, 56)]


if os.environ.get("WINO_ONLY"):
    PERF = [p_ for p_ in PERF if any(k in p_[0] for k in os.environ["WINO_ONLY"].split(","))]


def three(x, w, b, lrelu=False, qi=None, ref=None):
    """errors of (fp32-MFMA, direct fp16x2, Winograd fp16x2) against ref on the index qi"""
    qi = qi if qi is not None else (slice(None),)
    out = []
    xc, wc, bc = x.cuda(), w.cuda(), (b.cuda() if b is not None else None)
    for m in ("f32", "h2"):
        C.set_math(m)
        y = C.conv_forward(xc, wc, bc, 1, 1, lrelu)
        out.append(rel(y[qi], ref))
    C.set_math("h2")
    xa = C.amax_measure(xc)
    ya = C.Amax.zeros(xc.device, 1)
    y = wino_forward(xc, pack(wc), bc, lrelu, xa, ya)
    torch.cuda.synchronize()
    out.append(rel(y[qi], ref))
    fused = float(ya.slots[ya.first])
    assert fused == float(y.abs().max()), (fused, float(y.abs().max()))
    return out


def main():
    hip.lib().irr_conv_x3_set_min_blocks(0)
    worst = 0.0
    if "--noacc" not in sys.argv:
        print("== max|err|/max|ref| vs fp64: fp32-MFMA | direct fp16x2 | Winograd fp16x2   (Winograd / fp32-MFMA; bar 4x, floor 1e-6)")
        for rname, gen in RANGES.items():
            for cin, cout, B, H, W in ACC:
                g = torch.Generator().manual_seed(cin * 7 + cout)
                if gen is None:
                    x = torch.randn(B, cin, H, W, generator=g) * 1e-2
                    x[0, 0, 3, 3] = 1e4
                else:
                    x = gen(g, (B, cin, H, W))
                w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
                b = torch.randn(cout, generator=g) * 0.1 * float(x.abs().max()) * 1e-1
                ref = F.leaky_relu(F.conv2d(x.double(), w.double(), b.double(), padding=1), 0.1)
                e = three(x, w, b, True, None, ref)
                ok = e[2] <= max(4 * e[0], 1e-6)
                worst = max(worst, e[2] / max(e[0], 2.5e-7))
                print(f"{rname:24s} {cin:4d}->{cout:4d} {B}x{H}x{W}:  {e[0]:.2e} | {e[1]:.2e} | {e[2]:.2e}   ({e[2] / e[0]:.1f}x) {'ok' if ok else 'FAIL'}", flush=True)
        print("== regional: error of the QUIET half relative to its own range")
        for region in ("samples", "rows"):
            for ratio in (1e-5, 1e-6, 1e-7):
                for cin, cout, B, H, W in ACC[:4]:
                    B = max(B, 2)
                    g = torch.Generator().manual_seed(cin * 7 + cout)
                    x = torch.randn(B, cin, H, W, generator=g)
                    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
                    if region == "samples":
                        x[B // 2:] *= ratio
                        qi = (slice(B // 2, B),)
                    else:
                        x[:, :, H // 2:] *= ratio
                        qi = (slice(None), slice(None), slice(H // 2 + 1, H))
                    ref = F.conv2d(x.double(), w.double(), None, padding=1)[qi]
                    e = three(x, w, None, False, qi, ref)
                    ok = e[2] <= max(4 * e[0], 1e-6)
                    worst = max(worst, e[2] / max(e[0], 2.5e-7))
                    print(f"{region:8s} {ratio:.0e} {cin:4d}->{cout:4d} {B}x{H}x{W}:  {e[0]:.2e} | {e[1]:.2e} | {e[2]:.2e}   ({e[2] / e[0]:.1f}x) {'ok' if ok else 'FAIL'}", flush=True)
        print(f"worst Winograd / max(fp32-MFMA, 2.5e-7): {worst:.2f}x (bar 4x)")
    hip.lib().irr_conv_x3_set_min_blocks(384)
    if "--noperf" in sys.argv:
        return
    print("== speed: direct fp16x2 (conv_x3_kernel) vs Winograd fp16x2, forward + bias + LeakyReLU + fused amax; TFLOP/s of ALGORITHMIC fp32 work")
    C.set_math("h2")
    for name, cin, cout, B, H, W in PERF:
        x = torch.randn(B, cin, H, W, device="cuda")
        w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
        b = torch.randn(cout, device="cuda")
        gf = 2.0 * B * H * W * cout * cin * 9 / 1e9
        xa = C.amax_measure(x)
        ya = C.Amax.zeros(x.device, 1)
        pk = pack(w)
        y = torch.empty(B, cout, H, W, device="cuda")
        td = timeit(lambda: C.conv_forward(x, w, b, 1, 1, True, out=y, x_amax=xa, y_amax=ya))
        yd = y.clone()
        tw = timeit(lambda: wino_forward(x, pk, b, True, xa, ya, out=y))
        diff = float((y - yd).abs().max() / yd.abs().max())
        print(f"{name:26s} {cin:4d}->{cout:4d} {gf:7.1f} GF  direct {td:6.3f} ms {gf / td:6.1f} TF | winograd {tw:6.3f} ms {gf / tw:6.1f} TF | "
              f"{td / tw:4.2f}x  (max diff {diff:.1e})", flush=True)


if __name__ == "__main__":
    main()
