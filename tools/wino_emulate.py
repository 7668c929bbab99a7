"""Round 6, VERDICT r5 next #1 (gate, numerics side): HOST EMULATION of a Winograd F(2x2, 3x3) forward convolution on the fp16x2
split-operand arithmetic, next to the emulation of the direct fp16x2 kernel and an fp32 convolution, against fp64.

    U = G g G^T (fp32, per (co, ci); one power-of-two scale per matrix; plain fp16 pair  uh + ul)
    V = B^T d B (fp32 adds of the scaled input, |V| <= 4 max|x|: two more bits of head room; pair  vh + 2^-11 vl')
    M[xi] = sum_ci  uh vh + ul vh + (uh 2^-11) vl'      (fp32 accumulation, 16 GEMMs)
    Y = A^T M A (fp32)

Pieces are exact fp16 values held in fp32; piece products are exact in fp32; the sums run in fp32 (torch CPU matmul order -- not the
MFMA's order, the same error class).  Prints max|err| / max|ref| for the operand ranges and the regional cases of tests/test_h2_gpu.py."""
import sys
import torch
import torch.nn.functional as F

torch.set_num_threads(8)
Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def h2_exp(amax, head=0):
    if not (amax > 0):
        return 0
    import math
    m, e = math.frexp(float(amax))
    return 15 - e - head


def f16(t):
    return t.half().float()


def pair_up(t):
    """hi + 2^-11 lo' (low piece scaled up by 2^11: x3_split.h)"""
    hi = f16(t)
    lo = f16((t - hi) * 2048.0)
    return hi, lo


def pair_plain(t):
    hi = f16(t)
    return hi, f16(t - hi)


def conv_f32(x, w, dil=1):
    return F.conv2d(x, w, None, padding=dil, dilation=dil)


def conv_h2_direct(x, w):
    ex, ew = h2_exp(x.abs().max()), h2_exp(w.abs().max())
    xh, xl = pair_up(x * 2.0 ** ex)
    wh, wl = pair_plain(w * 2.0 ** ew)
    whd = f16(wh * 2.0 ** -11)
    acc = F.conv2d(xh, wl, None, padding=1) + F.conv2d(xl, whd, None, padding=1) + F.conv2d(xh, wh, None, padding=1)
    return acc * 2.0 ** -ex * 2.0 ** -ew


def tiles(x):
    """(B, C, H, W) -> (B, C, th, tw, 4, 4) input tiles of the 2x2 output tiles (zero padding 1, H and W even)"""
    B, C, H, W = x.shape
    xp = F.pad(x, (1, 1, 1, 1))
    return xp.unfold(2, 4, 2).unfold(3, 4, 2)


def conv_h2_wino(x, w):
    B, C, H, W = x.shape
    assert H % 2 == 0 and W % 2 == 0
    ex = h2_exp(x.abs().max(), head=2)
    d = tiles(x * 2.0 ** ex)                                       # exact scaling
    V = torch.einsum("ij,bcyxjk,lk->bcyxil", Bt, d, Bt)            # fp32 adds (exact +-: each entry is a sum of four terms)
    U = torch.einsum("ij,ocjk,lk->ocil", G, w, G)                  # fp32
    ew = h2_exp(U.abs().max())
    vh, vl = pair_up(V)
    uh, ul = pair_plain(U * 2.0 ** ew)
    uhd = f16(uh * 2.0 ** -11)
    M = (torch.einsum("ocil,bcyxil->boyxil", ul, vh) + torch.einsum("ocil,bcyxil->boyxil", uhd, vl)
         + torch.einsum("ocil,bcyxil->boyxil", uh, vh))
    Y = torch.einsum("pi,boyxil,ql->boyxpq", At, M, At)            # fp32
    Y = Y * 2.0 ** -ex * 2.0 ** -ew
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(B, w.shape[0], H, W)


def conv_f32_wino(x, w):
    """the same transforms with fp32 operands: what Winograd alone costs"""
    B, C, H, W = x.shape
    V = torch.einsum("ij,bcyxjk,lk->bcyxil", Bt, tiles(x), Bt)
    U = torch.einsum("ij,ocjk,lk->ocil", G, w, G)
    M = torch.einsum("ocil,bcyxil->boyxil", U, V)
    Y = torch.einsum("pi,boyxil,ql->boyxpq", At, M, At)
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(B, w.shape[0], H, W)


def rel(a, ref):
    return ((a.double() - ref).abs().max() / ref.abs().max()).item()


RANGES = {
    "unit": lambda g, s: torch.randn(s, generator=g),
    "per-channel 1e-2..1e2": lambda g, s: torch.randn(s, generator=g) * torch.exp(2.0 * torch.randn(s[0], s[1], 1, 1, generator=g)),
    "tiny (1e-7)": lambda g, s: torch.randn(s, generator=g) * 1e-7,
    "outlier 1e4": None,
    "relu-sparse": lambda g, s: torch.relu(torch.randn(s, generator=g)) * 3.0,
    "3e12": lambda g, s: torch.randn(s, generator=g) * 3e12,
}
CASES = [(115, 128, 2, 24, 28), (565, 128, 1, 16, 48), (243, 128, 2, 24, 28), (128, 64, 1, 40, 24)]


def main():
    print("== ranges: max|err|/max|ref| vs fp64:  fp32 direct | fp32 Winograd | h2 direct | h2 Winograd   (ratio h2-Winograd / fp32 direct)")
    for rname, gen in RANGES.items():
        for cin, cout, B, H, W in CASES:
            g = torch.Generator().manual_seed(cin * 7 + cout)
            if gen is None:
                x = torch.randn(B, cin, H, W, generator=g) * 1e-2
                x[0, 0, 3, 3] = 1e4
            else:
                x = gen(g, (B, cin, H, W))
            w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
            ref = F.conv2d(x.double(), w.double(), None, padding=1)
            e = [rel(conv_f32(x, w), ref), rel(conv_f32_wino(x, w), ref), rel(conv_h2_direct(x, w), ref), rel(conv_h2_wino(x, w), ref)]
            print(f"{rname:24s} {cin:4d}->{cout:4d} {B}x{H}x{W}:  {e[0]:.2e} | {e[1]:.2e} | {e[2]:.2e} | {e[3]:.2e}   ({e[3] / e[0]:.1f}x)", flush=True)
    print("== regional: error of the QUIET half relative to its own range")
    for region in ("samples", "rows"):
        for ratio in (1e-5, 1e-6, 1e-7, 1e-8):
            for cin, cout, B, H, W in CASES[:2] + CASES[3:]:
                B = max(B, 2)
                g = torch.Generator().manual_seed(cin * 7 + cout)
                x = torch.randn(B, cin, H, W, generator=g)
                w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
                if region == "samples":
                    x[B // 2:] *= ratio
                    qi = (slice(B // 2, B),)
                else:
                    x[:, :, H // 2:] *= ratio
                    qi = (slice(None), slice(None), slice(H // 2 + 1, H))
                ref = F.conv2d(x.double(), w.double(), None, padding=1)[qi]
                e = [rel(conv_f32(x, w)[qi], ref), rel(conv_f32_wino(x, w)[qi], ref), rel(conv_h2_direct(x, w)[qi], ref), rel(conv_h2_wino(x, w)[qi], ref)]
                print(f"{region:8s} {ratio:.0e} {cin:4d}->{cout:4d} {B}x{H}x{W}:  {e[0]:.2e} | {e[1]:.2e} | {e[2]:.2e} | {e[3]:.2e}   ({e[3] / e[0]:.1f}x)", flush=True)


if __name__ == "__main__":
    main()
