"""Round 5, NOTES D.2: WHAT differs when conv_x3s_kernel returns a wrong result beside a second process (tools/r5_concurrency_probe.py counts
such launches)?  Repeats one accumulate + mask data-gradient launch on fixed operands; for every launch that differs from the first,
lists the wrong elements by their role in the kernel's epilogue (a producer thread owns the quad q = (x % 32) / 4 of tile row y % 8 for
the eight channels of group c / 8: lane = q + 8 * row, wave = c / 8; px = x % 4 = component of its 16-B store) and how the wrong value
relates to the right one.   Run two copies at once:  python tools/r5_x3s_corruption_map.py 400 & python tools/r5_x3s_corruption_map.py 400"""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irr_amd import conv as C
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
C.set_math("h2"); C.set_x3s_h2(True)
torch.manual_seed(0)
B, H, W = 16, 224, 512
g = torch.randn(B, 32, H, W, device="cuda") * 1e-3
res = torch.randn(B, 32, H, W, device="cuda") * 1e-3
mask = torch.randn(B, 32, H, W, device="cuda")
w = torch.randn(32, 32, 3, 3, device="cuda") * 0.06
ga = C.amax_measure(g)
first = None
shown = 0
for it in range(N):
    gx = res.clone()
    C.conv_dgrad(g, w, 1, 1, (H, W), gx=gx, accumulate=True, mask=mask, nmask=32, gy_amax=ga)
    if first is None:
        first = gx
        continue
    if torch.equal(gx, first):
        continue
    idx = (gx != first).nonzero()
    b, c, y, x = idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]
    lane = ((x % 32) // 4 + 8 * (y % 8)).tolist()
    roles = collections.Counter(zip((c // 8).tolist(), (c % 8).tolist(), (x % 4).tolist()))
    lanes = collections.Counter(lane)
    tiles = collections.Counter(zip(b.tolist(), (y // 8).tolist(), (x // 32).tolist()))
    ratio = (gx[gx != first] / first[gx != first])
    # the epilogue computes v = (res + dgrad) * lrelu'(mask): is the wrong value the right one with ANOTHER mask factor / without res?
    m = torch.where(mask > 0, 1.0, 0.1)[gx != first]
    print(f"launch {it}: {idx.shape[0]} wrong elements in {len(tiles)} tile(s) {list(tiles.items())[:3]}; (wave, channel e, px) {sorted(roles.items())[:12]}; "
          f"lanes {sorted(lanes.items())}; wrong / right: {ratio[:8].tolist()}; mask factor of those elements {m[:8].tolist()}", flush=True)
    shown += 1
    if shown >= 6:
        break
print(f"done: {shown} differing launches shown of {N}", flush=True)
