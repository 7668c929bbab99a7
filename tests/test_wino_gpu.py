"""The Winograd F(2x2, 3x3) entry points on the fp16x2 arithmetic (csrc/conv_wino.hip, ABI 10; round 6, a gated experiment that did NOT
clear its speed gate -- profiles/NOTES.md E.1 -- and is not routed to): they are exported, so they are pinned: forward and the stride-1 data
gradient (transposed pack) against an fp64 convolution with the fp32-MFMA kernel's own error as the yardstick, ragged sizes, channel
tails, bias + LeakyReLU, the fused output magnitude, and the argument checks."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [(115, 128, 2, 24, 28), (64, 64, 1, 40, 24), (243, 96, 1, 30, 36), (128, 32, 2, 18, 44), (565, 128, 1, 16, 48)]   # (Cin, Cout, B, H, W)


def _pack(w, transpose=False):
    from irr_amd import hip
    cout, cin = (w.shape[1], w.shape[0]) if transpose else (w.shape[0], w.shape[1])
    uq = torch.empty(int(hip.lib().irr_conv_wino_packed_bytes(cin, cout)), dtype=torch.uint8, device=w.device)
    wmax = w.abs().max().reshape(1).float()
    hip.call("irr_conv_pack_weights_wino_h2", hip.ptr(w), uq.data_ptr(), cin, cout, int(transpose), hip.ptr(wmax), hip.stream())
    return uq, cin, cout


def _wino(x, packed, bias, lrelu, xa, ya=None):
    from irr_amd import hip
    uq, cin, cout = packed
    B, _, H, W = x.shape
    y = torch.empty(B, cout, H, W, device=x.device, dtype=torch.float32)
    hip.call("irr_conv2d_wino_fwd_h2", hip.ptr(x), uq.data_ptr(), hip.ptr(bias), hip.ptr(y), B, cin, H, W, cout, hip.bs(x), hip.bs(y),
             int(lrelu), 1.0, xa.ptr(), xa.n, ya.ptr() if ya is not None else None, hip.stream())
    return y


def _rel(a, ref):
    return ((a.cpu().double() - ref).abs().max() / ref.abs().max()).item()


@pytest.mark.parametrize("rng", ["unit", "outlier", "huge"])
@pytest.mark.parametrize("case", CASES, ids=[f"{c[0]}to{c[1]}_{c[2]}x{c[3]}x{c[4]}" for c in CASES])
def test_winograd_forward_and_data_gradient_are_fp32_faithful(case, rng):
    from irr_amd import conv as C, hip
    cin, cout, B, H, W = case
    if not hip.lib().irr_conv2d_wino_eligible(B, cin, H, W, cout):
        pytest.skip("shape outside the experimental launcher's range")
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(B, cin, H, W, generator=g)
    gy = torch.randn(B, cout, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    b = torch.linspace(-1, 1, cout) * 0.1
    if rng == "outlier":
        x = x * 1e-2
        x[0, 0, 3, 3] = 1e4
    elif rng == "huge":
        x, gy, w, b = x * 3e12, gy * 1e-20, w * 1e-6, b * 3e6
    ref = F.leaky_relu(F.conv2d(x.double(), w.double(), b.double(), padding=1), 0.1)
    gref = torch.nn.grad.conv2d_input(x.shape, w.double(), gy.double(), padding=1)
    xc, wc, bc, gc = x.cuda(), w.cuda(), b.cuda(), gy.cuda()
    old = C.MATH
    try:
        C.set_math("f32")
        e32 = (_rel(C.conv_forward(xc, wc, bc, 1, 1, True), ref), _rel(C.conv_dgrad(gc, wc, 1, 1, (H, W)), gref))
    finally:
        C.set_math(old)
    xa, ga, ya = C.amax_measure(xc), C.amax_measure(gc), C.Amax.zeros(xc.device, 1)
    y = _wino(xc, _pack(wc), bc, True, xa, ya)
    gx = _wino(gc, _pack(wc, transpose=True), None, False, ga)
    assert ya.slots[ya.first].item() == y.abs().max().item()
    for what, got, r, e in (("forward", y, ref, e32[0]), ("data gradient", gx, gref, e32[1])):
        err = _rel(got, r)
        assert err <= max(4 * e, 1e-6), (what, err, e)


def test_winograd_rejects_what_it_does_not_take():
    from irr_amd import conv as C, hip
    x = torch.randn(1, 64, 16, 18, device="cuda")                     # W % 4 != 0
    assert not hip.lib().irr_conv2d_wino_eligible(1, 64, 16, 18, 64)
    w = torch.randn(64, 64, 3, 3, device="cuda")
    xa = C.amax_measure(x)
    with pytest.raises(hip.HipError):
        _wino(x, _pack(w), None, False, xa)
