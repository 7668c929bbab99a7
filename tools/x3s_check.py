"""Streaming 32-channel conv kernel: epilogue combinations vs fp64, and speed at the OccUpsampleNetwork shapes."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irr_amd import conv as C, hip
from tools.x3_check import timeit


def main():

    hip.lib().irr_conv_x3_set_min_blocks(0)
    g = torch.Generator().manual_seed(3)
    for (cin, cout, B, H, W) in [(32, 32, 2, 24, 64), (32, 32, 1, 70, 90), (24, 32, 1, 16, 32), (32, 9, 2, 16, 96), (32, 32, 3, 33, 47)]:
        x = torch.randn(B, cin, H, W, generator=g); w = torch.randn(cout, cin, 3, 3, generator=g) * 0.1; b = torch.randn(cout, generator=g)
        res = torch.randn(B, cout, H, W, generator=g); base = torch.randn(B, cout, H, W, generator=g); msk = torch.randn(B, cout, H, W, generator=g)
        code = C.x3_code(B, cin, H, W, cout, 3, 1, 1)
        conv = F.conv2d(x.double(), w.double(), b.double(), padding=1)
        errs = []
        # 1: bias + lrelu
        y = C.conv_forward(x.cuda(), w.cuda(), b.cuda(), 1, 1, True)
        errs.append((y.cpu().double() - F.leaky_relu(conv, 0.1)).abs().max().item())
        # 2: res + alpha, no act
        y = C.conv_forward(x.cuda(), w.cuda(), b.cuda(), 1, 1, False, res=res.cuda(), alpha=0.1)
        errs.append((y.cpu().double() - (res.double() + 0.1 * conv)).abs().max().item())
        # 3: accumulate
        acc = base.clone().cuda()
        C.conv_forward(x.cuda(), w.cuda(), None, 1, 1, False, out=acc, accumulate=True)
        errs.append((acc.cpu().double() - (base.double() + F.conv2d(x.double(), w.double(), None, padding=1))).abs().max().item())
        # 4: dgrad with res + alpha + accumulate + mask (the OccUpsample i == 0 launch)
        gy = torch.randn(B, cout, H, W, generator=g)
        if C.x3_code(B, cout, H, W, cin, 3, 1, 1):
            gx = base[:, :cin].clone().cuda() if cin <= cout else torch.zeros(B, cin, H, W).cuda()
            b0 = gx.clone().cpu()
            r2 = torch.randn(B, cin, H, W, generator=g); m2 = torch.randn(B, cin, H, W, generator=g)
            C.conv_dgrad(gy.cuda(), w.cuda(), 1, 1, (H, W), gx=gx, accumulate=True, res=r2.cuda(), alpha=0.1, mask=m2.cuda(), nmask=cin)
            ref = (b0.double() + r2.double() + 0.1 * torch.nn.grad.conv2d_input(x.shape, w.double(), gy.double(), padding=1)) * torch.where(m2 > 0, 1.0, 0.1).double()
            errs.append((gx.cpu().double() - ref).abs().max().item())
        print(f"{cin}->{cout} {B}x{H}x{W} code {code}: " + " ".join(f"{e:.2e}" for e in errs), flush=True)
    hip.lib().irr_conv_x3_set_min_blocks(384)
    for name, cin, cout, B, H, W in [("occup L6", 32, 32, 64, 384, 448), ("occup L5", 32, 32, 64, 192, 224)]:
        x = torch.randn(B, cin, H, W, device="cuda"); w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05; b = torch.randn(cout, device="cuda")
        res = torch.randn(B, cout, H, W, device="cuda")
        gf = 2.0 * B * H * W * cout * cin * 9 / 1e9
        for m in ("f32", "x3"):
            C.set_math(m)
            t1 = timeit(lambda: C.conv_forward(x, w, b, 1, 1, True))
            t2 = timeit(lambda: C.conv_forward(x, w, b, 1, 1, False, res=res, alpha=0.1))
            gb = B * H * W * 4 * (cin + cout) / 1e9
            print(f"{name} {m}: plain {t1:6.2f} ms {gf / t1:6.1f} TF {gb / t1:5.2f} TB/s | +res {t2:6.2f} ms {(gb + B*H*W*4*cout/1e9) / t2:5.2f} TB/s  code {C.x3_code(B, cin, H, W, cout, 3, 1, 1)}", flush=True)


if __name__ == "__main__":
    main()
