import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irr_amd import conv as C
B, cin, cout, H, W = 64, 32, 32, 384, 448
x = torch.randn(B, cin, H, W, device="cuda"); w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05; b = torch.randn(cout, device="cuda")
for _ in range(4):
    C.conv_forward(x, w, b, 1, 1, True)
torch.cuda.synchronize()
