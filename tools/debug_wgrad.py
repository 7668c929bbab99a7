import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irr_amd import conv as C
import torch.nn.functional as F
torch.manual_seed(0)
for (B, cin, cout, H, W) in [(1, 11, 32, 48, 56), (2, 32, 32, 8, 60), (1, 64, 64, 6, 90), (2, 40, 128, 10, 64)]:
    x = torch.randn(B, cin, H, W, device="cuda"); gy = torch.randn(B, cout, H, W, device="cuda")
    gw = C.conv_wgrad(x, gy, (cout, cin, 3, 3), 1, 1)
    ref = torch.nn.grad.conv2d_weight(x.cpu(), (cout, cin, 3, 3), gy.cpu(), padding=1)
    d = (gw.cpu() - ref).abs()
    print((B, cin, cout, H, W), "max err", d.max().item(), "ref max", ref.abs().max().item())
    if d.max() > 1e-2:
        bad = (d > 1e-2).nonzero()
        print("  bad count", len(bad), "of", d.numel(), "first", bad[:6].tolist(), "taps bad:", sorted(set((b[2].item(), b[3].item()) for b in bad)))
        print("  bad co:", sorted(set(b[0].item() for b in bad))[:40])
        print("  bad ci:", sorted(set(b[1].item() for b in bad))[:40])
