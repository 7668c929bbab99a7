"""Is conv_x3 clock/power limited?  Same launch on random data, on all-zero data and on constant data."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irr_amd import conv as C
from tools.x3_check import timeit
B, cin, cout, H, W = 64, 565, 128, 96, 112
gf = 2.0 * B * H * W * cout * cin * 9 / 1e9
for name, fx, fw in (("random", torch.randn, torch.randn), ("zeros", torch.zeros, torch.zeros), ("ones", torch.ones, torch.ones)):
    x = fx(B, cin, H, W, device="cuda"); w = fw(cout, cin, 3, 3, device="cuda") * 0.05; b = torch.zeros(cout, device="cuda")
    for m in ("f32", "x3"):
        C.set_math(m)
        t = timeit(lambda: C.conv_forward(x, w, b, 1, 1, True), iters=10)
        print(f"{name:8s} {m}: {t:6.2f} ms {gf / t:6.1f} TF", flush=True)
