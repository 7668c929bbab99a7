"""Round 5: the bit-mask launches of the streaming kernel (irr_conv2d_fwd_h2_bits) beside the weight-gradient lane's kernels.
bench.py at bs32 with the lane on ended in non-finite gradients in step 2 with the bit masks, never without the lane / at bs8 / with
fp32 masks.  Victims: the plain forward that WRITES bits (y and the words compared), the masked data gradient that READS them, the
same gradient with the fp32 mask; aggressors on a second stream: the 32 -> 32 weight gradient of the same maps (the lane's kernel at
that level), the dilation-16 one of NOTES C.3.  Bit-compared with the lone launch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irr_amd import conv as C

torch.manual_seed(0)
B, H, W = int(os.environ.get("PB", "16")), 384, 448
C.set_math("h2")
side = torch.cuda.Stream()
x = torch.randn(B, 32, H, W, device="cuda")
w = torch.randn(32, 32, 3, 3, device="cuda") * 0.05
bias = torch.randn(32, device="cuda") * 0.05
gy = torch.randn(B, 32, H, W, device="cuda") * 1e-4
res = torch.randn(B, 32, H, W, device="cuda") * 1e-4
xa, ga = C.amax_measure(x), C.amax_measure(gy)
nw = C.x3s_mask_words(B, H, W)
assert C.x3s_bits_ok(B, 32, H, W, 32)
x16 = torch.randn(8, 96, 96, 112, device="cuda")
g16 = torch.randn(8, 64, 96, 112, device="cuda") * 1e-6
gw16 = torch.zeros(64, 96, 3, 3, device="cuda")
gw32 = torch.zeros(32, 32, 3, 3, device="cuda")
x16a, g16a = C.amax_measure(x16), C.amax_measure(g16)
bits0 = torch.empty(nw, dtype=torch.int32, device="cuda")
y0 = C.conv_forward(x, w, bias, 1, 1, True, x_amax=xa, bits_out=bits0)
torch.cuda.synchronize()


def victim(kind):
    if kind == "fwd+bits":
        b = torch.zeros(nw, dtype=torch.int32, device="cuda")
        y = C.conv_forward(x, w, bias, 1, 1, True, x_amax=xa, bits_out=b)
        return (y, b)
    if kind == "fwd":
        return (C.conv_forward(x, w, bias, 1, 1, True, x_amax=xa),)
    if kind == "dgrad bits":
        return (C.conv_dgrad(gy, w, 1, 1, (H, W), mask=y0, nmask=32, res=res, alpha=0.1, gy_amax=ga, mask_bits=bits0),)
    if kind == "dgrad fp32 mask":
        return (C.conv_dgrad(gy, w, 1, 1, (H, W), mask=y0, nmask=32, res=res, alpha=0.1, gy_amax=ga),)
    raise SystemExit(kind)


def aggressor(kind):
    if kind == "wgrad 32->32":
        for _ in range(2):
            C.conv_wgrad(x, gy, gw32.shape, 1, 1, gw=gw32, x_amax=xa, gy_amax=ga)
    elif kind == "wgrad d16":
        for _ in range(6):
            C.conv_wgrad(x16, g16, gw16.shape, 1, 16, gw=gw16, x_amax=x16a, gy_amax=g16a)


for vk in ("fwd+bits", "fwd", "dgrad bits", "dgrad fp32 mask"):
    ref = victim(vk)
    torch.cuda.synchronize()
    for ak in ("none", "wgrad 32->32", "wgrad d16"):
        bad, detail = 0, ""
        for rep in range(24):
            if ak != "none":
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    aggressor(ak)
                if rep % 4:
                    torch.cuda._sleep(200000 * (rep % 4))
            out = victim(vk)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            ok = all(torch.equal(a, b) for a, b in zip(out, ref))
            if not ok:
                bad += 1
                if not detail:
                    d = [(int((a != b).sum()), float((a.double() - b.double()).abs().max())) for a, b in zip(out, ref)]
                    nf = [int((~torch.isfinite(a)).sum()) if a.dtype == torch.float32 else 0 for a in out]
                    detail = f" first: (elements differing, max |diff|) {d}, non-finite {nf}"
        print(f"{vk:16s} beside {ak:13s}: {bad} of 24 differ from the lone launch{detail}", flush=True)
