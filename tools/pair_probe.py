"""Round 4, NOTES C.3: the launch pair behind the lane deviation.  A victim launch (VICTIM=dgrad: the Cout = 2 data gradient of conv_last,
accumulate + mask over a 563-channel buffer; dgrad_plain, fwd_smallco, x3s, conv115, torch) on the main stream while the dilation-16
weight gradient runs on a second stream: is the victim's result bit-equal to the lone launch?  With packed fp32 instructions in the
victim (library built WITH the vectorisers) 30-39 of 40 results differ by 4e-3; built without them none does."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irr_amd import conv as C, hip

torch.manual_seed(0)
B, H, W = 8, 96, 112
side = torch.cuda.Stream()
g_est = torch.randn(B, 2, H, W, device="cuda") * 1e-5
w_last = torch.randn(2, 563, 3, 3, device="cuda") * 0.02
G0 = torch.randn(B, 565, H, W, device="cuda") * 1e-6
buf = torch.randn(B, 565, H, W, device="cuda")
x16 = torch.randn(B, 96, H, W, device="cuda")
g16 = torch.randn(B, 64, H, W, device="cuda") * 1e-6
gw16 = torch.zeros(64, 96, 3, 3, device="cuda")
x2 = torch.randn(B, 128, H, W, device="cuda")
g2 = torch.randn(B, 128, H, W, device="cuda") * 1e-6
gw2 = torch.zeros(128, 128, 3, 3, device="cuda")


VICTIM = os.environ.get("VICTIM", "dgrad")
xs32 = torch.randn(B, 32, H, W, device="cuda")
w32 = torch.randn(32, 32, 3, 3, device="cuda") * 0.05
w128 = torch.randn(128, 115, 3, 3, device="cuda") * 0.05
b2 = torch.zeros(2, device="cuda")


def victim():
    if VICTIM == "dgrad":
        G = G0.clone()
        C.conv_dgrad(g_est, w_last, 1, 1, (H, W), gx=G[:, :563], accumulate=True, mask=buf[:, :563], nmask=32)
        return G
    if VICTIM == "dgrad_plain":
        return C.conv_dgrad(g_est, w_last, 1, 1, (H, W))
    if VICTIM == "fwd_smallco":
        return C.conv_forward(buf[:, :563], w_last, b2, 1, 1, False)
    if VICTIM == "x3s":
        return C.conv_forward(xs32, w32, None, 1, 1, True)
    if VICTIM == "conv115":
        return C.conv_forward(buf[:, :115], w128, None, 1, 1, True)
    if VICTIM == "torch":
        return buf * 1.5 + G0
    raise SystemExit(VICTIM)


ref = victim()
torch.cuda.synchronize()
for name, dil, xs, gs, gws in (("none", 0, None, None, None), ("d16 wgrad", 16, x16, g16, gw16)):
    for math in ("h2",):
        C.set_math(math)
        bad = 0
        worst = 0.0
        for rep in range(40):
            if dil:
                xa, ga = (C.amax_measure(xs), C.amax_measure(gs)) if math == "h2" else (None, None)
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(3):
                        C.conv_wgrad(xs, gs, gws.shape, 1, dil, gw=gws, x_amax=xa, gy_amax=ga)
                if rep % 4:
                    torch.cuda._sleep(20000 * (rep % 4))
            out = victim()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            if not torch.equal(out, ref):
                bad += 1
                worst = max(worst, ((out - ref).double().norm() / ref.double().norm()).item())
        print(f"beside {name} ({math}): {bad} of 40 results differ from the lone launch (worst {worst:.1e})", flush=True)
