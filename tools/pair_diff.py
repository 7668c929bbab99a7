"""Round 4, NOTES C.3: WHAT differs when the victim of tools/pair_probe.py deviates (run with IRR_HIP_LIB = a library whose
conv_small.hip was built with the vectorisers: tools/build_variant.py slp).  Prints, for the first deviating launches, how many
elements differ, where (batch / channel / row / column, thread-quad and wave alignment), by how much, and whether the wrong value
equals the FMA chain of that output with the accumulator zeroed after m products (bit for bit).

Result (profiles/r4_pair_diff.txt): always 16 lanes -- 48..63, the last quarter pass of a wave -- of the fourth pixel of the fourth
channel of a loop trip, and always m = 8: the ninth product is the one v_pk_fma_f32 of the chain whose accumulator halves are
swapped (op_sel:[0,0,1] op_sel_hi:[0,1,0]); its low result is computed with a zero accumulator.  tools/pkfma_swap.py replays that
instruction stand-alone."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irr_amd import conv as C

torch.manual_seed(0)
B, H, W = 8, 96, 112
side = torch.cuda.Stream()
g_est = torch.randn(B, 2, H, W, device="cuda") * 1e-5
w_last = torch.randn(2, 563, 3, 3, device="cuda") * 0.02
x16 = torch.randn(B, 96, H, W, device="cuda")
g16 = torch.randn(B, 64, H, W, device="cuda") * 1e-6
gw16 = torch.zeros(64, 96, 3, 3, device="cuda")
C.set_math("h2")


def victim():
    return C.conv_dgrad(g_est, w_last, 1, 1, (H, W))


ref = victim()
torch.cuda.synchronize()
truth = torch.nn.functional.conv_transpose2d(g_est.double(), w_last.double(), padding=1)
print(f"lone launch vs fp64: {((ref.double() - truth).norm() / truth.norm()).item():.2e}")
shown = bad = 0
for rep in range(40):
    xa, ga = C.amax_measure(x16), C.amax_measure(g16)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            C.conv_wgrad(x16, g16, gw16.shape, 1, 16, gw=gw16, x_amax=xa, gy_amax=ga)
    if rep % 4:
        torch.cuda._sleep(20000 * (rep % 4))
    out = victim()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    if torch.equal(out, ref):
        continue
    bad += 1
    if shown >= 4:
        continue
    shown += 1
    ne = (out != ref)
    idx = ne.nonzero()
    n = idx.shape[0]
    print(f"rep {rep}: {n} of {ref.numel()} elements differ; NaN/inf in out: {(~torch.isfinite(out)).sum().item()}")
    b, c, y, x = idx.unbind(1)
    print("  batches", torch.unique(b).tolist(), " channels: n =", torch.unique(c).numel(), "first", torch.unique(c)[:24].tolist())
    print("  rows", torch.unique(y)[:16].tolist(), " columns mod 4 histogram", torch.bincount(x % 4, minlength=4).tolist())
    quad = (y * W + x) // 4                       # thread index within the plane (a thread owns four adjacent pixels)
    print("  thread quads: n =", torch.unique(quad).numel(), " waves (quad // 64):", torch.unique(quad // 64)[:16].tolist(),
          " blocks (quad // 256):", torch.unique(quad // 256)[:16].tolist())
    print("  channels mod 4 histogram", torch.bincount(c % 4, minlength=4).tolist(), " (batches of four channels per loop trip)")
    o, r = out[ne].double(), ref[ne].double()
    rel = ((o - r).abs() / r.abs().clamp_min(1e-30))
    print(f"  |out-ref|/|ref|: median {rel.median().item():.2e}  max {rel.max().item():.2e}  min {rel.min().item():.2e};  out == 0: {(o == 0).sum().item()}")
    # quarter-wave structure: lanes of a wave that differ, per (wave, channel)
    lane = quad % 64
    key = (b * 1000 + c) * 100000 + quad // 64
    uk, cnt = torch.unique(key, return_counts=True)
    print("  differing lanes per (batch, channel, wave): histogram of counts", torch.unique(cnt, return_counts=True))
    k0 = key == uk[0]
    print("  lanes of the first such wave:", sorted(lane[k0].tolist()))
    # a STALE accumulator?  The accumulator registers of channel ci are the destination of a 16-byte buffer load issued at the top of
    # the loop trip (zeros here: no accumulate) and, before that, held the result of channel ci - 4.  Emulate the fp32 FMA chain
    # (taps in program order; a product of two fp32 values is exact in fp64) from a start value and compare bit for bit.
    import numpy as np
    def chain(bb, ci, yy, xx, start):
        acc = np.float32(start)
        for cch in range(2):
            for a in range(3):
                for t in range(3):
                    iy, ix = yy + 1 - a, xx + 1 - t
                    gv = g_est[bb, cch, iy, ix].item() if 0 <= iy < H and 0 <= ix < W else 0.0
                    acc = np.float32(np.float64(w_last[cch, ci, a, t].item()) * np.float64(gv) + np.float64(acc))
        return acc
    def chain_reset(bb, ci, yy, xx, m):
        """the chain whose accumulator is overwritten with 0 after the first m products"""
        acc = np.float32(0.0)
        q = 0
        for cch in range(2):
            for a in range(3):
                for t in range(3):
                    if q == m:
                        acc = np.float32(0.0)
                    iy, ix = yy + 1 - a, xx + 1 - t
                    gv = g_est[bb, cch, iy, ix].item() if 0 <= iy < H and 0 <= ix < W else 0.0
                    acc = np.float32(np.float64(w_last[cch, ci, a, t].item()) * np.float64(gv) + np.float64(acc))
                    q += 1
        return acc
    tally = {}
    for j in range(min(n, 96)):
        bb, ci, yy, xx = idx[j].tolist()
        o = np.float32(out[bb, ci, yy, xx].item())
        assert chain(bb, ci, yy, xx, 0.0) == np.float32(ref[bb, ci, yy, xx].item()), "the emulated chain does not reproduce the lone launch"
        found = "none"
        for m in range(1, 18):
            if chain_reset(bb, ci, yy, xx, m) == o:
                found = f"accumulator zeroed after {m} products"
        tally[found] = tally.get(found, 0) + 1
    print("  emulated chain, first", min(n, 96), "differing elements:", tally)
print(f"{bad} of 40 launches deviate")
