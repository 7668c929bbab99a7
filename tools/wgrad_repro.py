"""Bit-reproducibility of conv_wgrad over repeated launches (x3 kernels: partial slots summed in fixed order -> identical bits;
code 0 = fp32 kernels with an atomic flush -> last-bit differences) and the error against an fp64 reference."""
import sys, torch
sys.path.insert(0, "/root/repo")
from irr_amd import conv as C, hip
torch.manual_seed(0)
for (cin, cout, B, H, W) in ((531, 32, 8, 48, 56), (531, 32, 8, 96, 112), (531, 32, 8, 24, 64), (300, 32, 4, 48, 56), (64, 9, 8, 48, 56), (531, 32, 64, 48, 56)):
    x = torch.randn(B, cin, H, W, device="cuda"); gy = torch.randn(B, cout, H, W, device="cuda")
    outs = []
    for _ in range(3):
        gw = torch.zeros(cout, cin, 3, 3, device="cuda")
        C.conv_wgrad(x, gy, (cout, cin, 3, 3), 1, 1, gw=gw)
        outs.append(gw.clone())
    ref = torch.nn.grad.conv2d_weight(x.double(), (cout, cin, 3, 3), gy.double(), padding=1)
    d01 = (outs[0] - outs[1]).abs().max().item(); d12 = (outs[1] - outs[2]).abs().max().item()
    err = [(o.double() - ref).abs().max().item() / ref.abs().max().item() for o in outs]
    nd = (outs[0] != outs[1]).sum().item()
    print(cin, cout, B, H, W, "code", hip.lib().irr_conv2d_wgrad_x3_eligible(B, cin, H, W, cout, 3, 1, 1), "diff", d01, d12, "n differing", nd, "err vs fp64", err)
