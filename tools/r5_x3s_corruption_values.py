"""Round 5, NOTES D.2: WHICH intermediate value ends up in the corrupted lanes of conv_x3s_kernel beside a second process?  Plain forward
launch (EPI 0) on constant input 1.0 with identity weights, bias_e = 10 + e', LeakyReLU on, alpha = 3 -- every stage of the epilogue has
a distinct value per channel co: accumulator 1, + bias 11 + co, 0.1 x that, result 3 * (11 + co).  Run two copies at once."""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irr_amd import conv as C
N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
C.set_math("h2"); C.set_x3s_h2(True)
B, H, W = 16, 224, 512
x = torch.ones(B, 32, H, W, device="cuda")
w = torch.zeros(32, 32, 3, 3, device="cuda")
for c in range(32):
    w[c, c, 1, 1] = 1.0
bias = torch.arange(32, device="cuda", dtype=torch.float32) + 10.0
xa = C.amax_measure(x)
want = (3.0 * (11.0 + torch.arange(32, device="cuda", dtype=torch.float32))).view(1, 32, 1, 1)
shown = 0
for it in range(N):
    out = torch.empty(B, 32, H, W, device="cuda")
    C.conv_forward(x, w, bias, 1, 1, True, out=out, alpha=3.0, x_amax=xa)
    badm = out != want
    if not bool(badm.any()):
        continue
    idx = badm.nonzero()
    c, y, xx = idx[:, 1], idx[:, 2], idx[:, 3]
    lanes = sorted(collections.Counter(((xx % 32) // 4 + 8 * (y % 8)).tolist()).items())
    print(f"launch {it}: {idx.shape[0]} wrong; channels {sorted(set(c.tolist()))} px {sorted(set((xx % 4).tolist()))} lanes {lanes}; values {out[badm][:8].tolist()} "
          f"(right: {want.expand_as(out)[badm][:8].tolist()})", flush=True)
    shown += 1
    if shown >= 8:
        break
print(f"done: {shown} wrong launches of {N}", flush=True)
