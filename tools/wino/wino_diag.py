"""where does conv_wino_kernel differ from the direct fp16x2 kernel?  (diagnosis)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irr_amd import conv as C, hip
from tools.wino_check import pack, wino_forward
B, cin, cout, H, W = [int(v) for v in os.environ.get("SHAPE", "4,32,64,96,112").split(",")]
torch.manual_seed(0)
x = torch.randn(B, cin, H, W, device="cuda"); w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
C.set_math("h2")
hip.lib().irr_conv_x3_set_min_blocks(0)
xa = C.amax_measure(x)
yd = C.conv_forward(x, w, None, 1, 1, False, x_amax=xa)
def show(d):
    bad = d > 1e-3
    print("  per sample:", bad.flatten(1).sum(1).tolist())
    print("  per co-tile of 32:", bad.sum((0, 2, 3)).view(-1, 32).sum(1).tolist())
    print("  per 16-row band:", bad.sum((0, 1, 3)).view(-1, 16).sum(1).tolist() if H % 16 == 0 else "-")
    print("  per 16-col band:", bad.sum((0, 1, 2))[: (W // 16) * 16].view(-1, 16).sum(1).tolist())
    print("  per row in tile (y%16):", bad.sum((0, 1, 3)).view(-1, 16).sum(0).tolist() if H % 16 == 0 else "-")
    print("  per col in tile (x%16):", bad.sum((0, 1, 2))[: (W // 16) * 16].view(-1, 16).sum(0).tolist())
    idx = bad.nonzero()
    print("  first wrong:", idx[:6].tolist(), "values", [(float(y[tuple(i)]), float(yd[tuple(i)])) for i in idx[:3]])


pk = pack(w)
for rep in range(int(os.environ.get("REPS", 4))):
    if rep == 2:
        pk = pack(w)
    y = wino_forward(x, pk, None, False, xa)
    torch.cuda.synchronize()
    d = (y - yd).abs()
    print(f"rep {rep}: max diff {float(d.max()):.3e} (max |y| {float(yd.abs().max()):.3f}); wrong elements {int((d > 1e-3).sum())} of {d.numel()}")
    if (d > 1e-3).any():
        show(d)
