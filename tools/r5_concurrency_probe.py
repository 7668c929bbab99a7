"""Round 5, NOTES D.2: do the fp16x2 launches of the OccUpsampleNetwork backward return the same bits while ANOTHER PROCESS keeps the GPU
busy?  The streaming kernel's data gradient (accumulate + mask epilogue, fused output magnitude) and the 32 -> 32 weight gradient on the
fp16x2 form are launched N times on fixed operands with a gradient-like dynamic range (most low pieces of the PLAIN pair are fp16
subnormals); every result is compared bit for bit with the first one, non-finite values are counted.

    python tools/r5_concurrency_probe.py [N]                 alone
    (python bench.py --steps 40 --no-cpu-baseline ... &)     ... and beside a second process (tools/r5_concurrency_probe.sh)
IRR_HIP_LIB=<plain-low-piece build> selects round 4's arithmetic."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irr_amd import conv as C

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
C.set_math("h2")
C.set_x3s_h2(True)
torch.manual_seed(0)
B, H, W = 16, 224, 512
g = torch.randn(B, 32, H, W, device="cuda") * torch.exp(3.0 * torch.randn(B, 32, H, W, device="cuda")) * 1e-4
x = torch.nn.functional.leaky_relu(torch.randn(B, 32, H, W, device="cuda"), 0.1)
res = torch.randn(B, 32, H, W, device="cuda") * 1e-4
mask = torch.randn(B, 32, H, W, device="cuda")
w = torch.nn.Parameter(torch.randn(32, 32, 3, 3, device="cuda") * 0.06)
assert C.h2_code(B, 32, H, W, 32, 3, 1, 1) == 9001
ga = C.amax_measure(g)
xa = C.amax_measure(x)
first = None
bad = {"dgrad": 0, "wgrad": 0, "amax": 0, "nonfinite": 0}
for it in range(N):
    gx = res.clone()
    sl = C.Amax.zeros(g.device, 1)
    C.conv_dgrad(g, w, 1, 1, (H, W), gx=gx, accumulate=True, mask=mask, nmask=32, gy_amax=ga, gx_amax=sl)
    gw = C.conv_wgrad(x, g, w.shape, 1, 1, x_amax=xa, gy_amax=ga)
    cur = (gx, gw, sl.slots[sl.first].clone())
    if first is None:
        first = cur
        torch.cuda.synchronize()
        print("reference launch: max|gx| %.4e (slot %.4e), max|gw| %.4e" % (float(gx.abs().max()), float(cur[2]), float(gw.abs().max())), flush=True)
        continue
    bad["dgrad"] += int(not torch.equal(cur[0], first[0]))
    bad["wgrad"] += int((cur[1] - first[1]).abs().max() > 1e-4 * first[1].abs().max())      # (the weight gradient's bias atomics are not part of it)
    bad["amax"] += int(float(cur[2]) != float(first[2]))
    bad["nonfinite"] += int(not (bool(torch.isfinite(cur[0]).all()) and bool(torch.isfinite(cur[1]).all())))
torch.cuda.synchronize()
print(f"{N - 1} repeated launches: {bad}  (library: {os.environ.get('IRR_HIP_LIB', 'product')})", flush=True)
