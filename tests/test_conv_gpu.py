"""GPU parity of the fp32-MFMA conv family against torch's CPU convolution (what the reference executes:
nn.Conv2d + LeakyReLU, models/pwc_modules.py:8-19), forward and all three gradients."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (Cin, Cout, k, stride, dil, lrelu, B, H, W)  -- every (k, stride, dil) and the awkward channel counts of Appendix A
CASES = [
    (3, 16, 3, 2, 1, True, 2, 32, 48),
    (16, 16, 3, 1, 1, True, 2, 16, 24),
    (128, 196, 3, 2, 1, True, 2, 12, 14),
    (115, 128, 3, 1, 1, True, 1, 24, 28),
    (243, 128, 3, 1, 1, True, 1, 12, 14),
    (565, 128, 3, 1, 1, True, 1, 12, 14),
    (128, 128, 3, 1, 2, True, 1, 24, 28),
    (128, 96, 3, 1, 8, True, 1, 24, 28),
    (96, 64, 3, 1, 16, True, 2, 24, 28),
    (64, 32, 3, 1, 1, True, 1, 24, 28),
    (32, 2, 3, 1, 1, False, 2, 24, 28),
    (562, 1, 3, 1, 1, False, 1, 12, 14),
    (32, 9, 3, 1, 1, True, 2, 12, 14),
    (196, 32, 1, 1, 1, True, 2, 6, 7),
    (16, 3, 1, 1, 1, True, 1, 48, 56),
    (11, 32, 3, 1, 1, True, 1, 48, 56),
    (32, 32, 3, 1, 1, False, 3, 33, 47),      # ragged: pixel count not a multiple of the tile
    (40, 128, 3, 1, 1, True, 2, 10, 64),      # halo-tile wgrad: partial last column tile + padded channels
    (64, 64, 3, 1, 1, True, 1, 7, 90),        # halo-tile wgrad: odd row count
    (35, 96, 3, 1, 1, True, 1, 12, 58),
    (3, 16, 3, 2, 1, True, 3, 30, 44),       # first pyramid conv: three-input-channel weight-gradient kernel
    (3, 18, 3, 1, 1, False, 2, 17, 23),
    (3, 16, 3, 2, 1, True, 2, 33, 47),       # odd sizes: the 2x2-block image-gradient kernel with ragged last row / column
    (2, 20, 3, 2, 1, False, 1, 21, 20),
    (4, 8, 3, 2, 1, True, 2, 16, 31),
    (18, 40, 3, 2, 1, True, 2, 20, 26),      # stride-2 data gradient in channel groups of four (ragged last group)
    (64, 96, 3, 2, 1, True, 1, 12, 14),
]


def _ref(x, w, b, k, stride, dil, lrelu):
    y = F.conv2d(x, w, b, stride=stride, padding=((k - 1) * dil) // 2, dilation=dil)
    return F.leaky_relu(y, 0.1) if lrelu else y


@pytest.mark.parametrize("case", CASES, ids=[f"{c[0]}to{c[1]}k{c[2]}s{c[3]}d{c[4]}" for c in CASES])
def test_conv_block_fwd_bwd(case):
    from irr_amd import conv as C
    cin, cout, k, stride, dil, lrelu, B, H, W = case
    g = torch.Generator().manual_seed(cin * 1000 + cout)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    xc, wc, bc = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = _ref(xc, wc, bc, k, stride, dil, lrelu)
    go = torch.randn(yr.shape, generator=g)
    yr.backward(go)
    xd, wd, bd = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    y = C.conv_block(xd, wd, bd, stride, dil, lrelu)
    y.backward(go.cuda())
    tol = dict(rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(y.detach().cpu().numpy(), yr.detach().numpy(), **tol)
    np.testing.assert_allclose(xd.grad.cpu().numpy(), xc.grad.numpy(), **tol)
    scale = float(wc.grad.abs().max())
    np.testing.assert_allclose(wd.grad.cpu().numpy(), wc.grad.numpy(), rtol=1e-3, atol=2e-4 * max(scale, 1.0))
    np.testing.assert_allclose(bd.grad.cpu().numpy(), bc.grad.numpy(), rtol=1e-3, atol=1e-3)


def test_conv_channel_slice_residual_accumulate():
    """DenseNet-style use: input is a channel SUFFIX of a buffer, output a slice of the same buffer;
    residual epilogue and accumulate-into-output for gradient buffers."""
    from irr_amd import conv as C
    g = torch.Generator().manual_seed(5)
    buf = torch.randn(2, 50, 20, 24, generator=g)
    w = torch.randn(12, 30, 3, 3, generator=g) * 0.1
    b = torch.randn(12, generator=g) * 0.1
    res = torch.randn(2, 12, 20, 24, generator=g)
    ref = res + 0.1 * _ref(buf[:, 20:], w, b, 3, 1, 1, True)
    d = buf.cuda()
    out = C.conv_forward(d[:, 20:], w.cuda(), b.cuda(), 1, 1, True, out=d[:, 8:20], res=res.cuda(), alpha=0.1)
    assert out.data_ptr() == d[:, 8:20].data_ptr()
    np.testing.assert_allclose(d[:, 8:20].cpu().numpy(), ref.numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(d[:, 20:].cpu().numpy(), buf[:, 20:].numpy())       # input untouched
    np.testing.assert_allclose(d[:, :8].cpu().numpy(), buf[:, :8].numpy())
    # accumulate
    base = torch.randn(2, 12, 20, 24, generator=g)
    acc = base.cuda()
    C.conv_forward(d[:, 20:], w.cuda(), None, 1, 1, False, out=acc, accumulate=True)
    ref2 = base + F.conv2d(buf[:, 20:], w, None, padding=1)
    np.testing.assert_allclose(acc.cpu().numpy(), ref2.numpy(), rtol=2e-4, atol=2e-4)


def test_conv_level4_shape_linearity():
    """BASELINE-size property check (bs8 level-4 decoder conv): conv(a*x) == a*conv(x) without bias/act,
    and equality with the MIOpen result of the same op on the same device."""
    from irr_amd import conv as C
    torch.manual_seed(0)
    x = torch.randn(8, 115, 96, 112, device="cuda")
    w = torch.randn(128, 115, 3, 3, device="cuda") * 0.03
    y1 = C.conv_forward(x, w, None, 1, 1, False)
    y2 = C.conv_forward(2 * x, w, None, 1, 1, False)
    assert torch.allclose(2 * y1, y2, rtol=1e-5, atol=1e-5)
    yr = F.conv2d(x, w, None, padding=1)
    assert (y1 - yr).abs().max().item() < 2e-3 * yr.abs().max().item()


# ---- conv_x3: fp32-faithful convolution on the bf16 matrix pipe (csrc/conv_x3.hip) -------------------------------
X3_CASES = [  # (Cin, Cout, dil, B, H, W): every block shape (CT=4/3/2/1), NT=7/8 tiles, both LDS planes, ragged edges, channel tails
    (115, 128, 1, 2, 24, 28), (565, 128, 1, 1, 16, 48), (371, 96, 1, 1, 33, 47), (531, 32, 1, 1, 32, 48),
    (128, 64, 1, 1, 40, 24), (64, 32, 1, 1, 70, 90), (128, 128, 2, 1, 24, 28), (128, 128, 4, 1, 24, 28),
    (16, 565, 1, 1, 24, 28), (243, 128, 1, 1, 48, 56), (35, 96, 1, 1, 12, 58),
    # row-folded dilated patches (rows y = r mod dil form one block row set): dilation 8 / 16, heights not divisible by dil
    (128, 96, 8, 1, 50, 56), (96, 128, 8, 1, 48, 56), (64, 96, 16, 1, 90, 112), (128, 128, 2, 1, 25, 28), (128, 128, 4, 2, 30, 36),
    (16, 16, 1, 2, 24, 64),                               # 16 -> 16 pyramid convs: two chunks, the second with zero weights
]


@pytest.fixture
def x3_everywhere():
    from irr_amd import conv as C, hip
    old = hip.lib().irr_conv_x3_set_min_blocks(0)
    C.set_math("x3")
    yield
    hip.lib().irr_conv_x3_set_min_blocks(old)
    C.set_math(C.DEFAULT_MATH)


@pytest.mark.parametrize("case", X3_CASES, ids=[f"{c[0]}to{c[1]}d{c[2]}_{c[4]}x{c[5]}" for c in X3_CASES])
def test_conv_x3_is_fp32_faithful(case, x3_everywhere):
    """Forward, data gradient (transposed pack) and the fused epilogue of conv_x3 against an fp64 convolution:
    the error must stay in the fp32 class (<= 4x the error of the fp32-MFMA kernel and <= 5e-6 of the output range;
    a single-bf16 product would be ~4e-3, tf32 ~5e-4)."""
    from irr_amd import conv as C
    cin, cout, dil, B, H, W = case
    g = torch.Generator().manual_seed(cin * 7 + cout)
    x = torch.randn(B, cin, H, W, generator=g) * torch.exp(torch.randn(B, cin, 1, 1, generator=g))   # per-channel scales
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    gy = torch.randn(B, cout, H, W, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=dil, dilation=dil)
    gref = torch.nn.grad.conv2d_input(x.shape, w.double(), gy.double(), padding=dil, dilation=dil)
    assert C.x3_code(B, cin, H, W, cout, 3, 1, dil) != 0, "case must be routed to conv_x3"
    err = {}
    for m in ("f32", "x3"):
        C.set_math(m)
        y = C.conv_forward(x.cuda(), w.cuda(), b.cuda(), 1, dil, False)
        e_f = (y.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
        e_d = None
        if m == "f32" or C.x3_code(B, cout, H, W, cin, 3, 1, dil):
            gx = C.conv_dgrad(gy.cuda(), w.cuda(), 1, dil, (H, W))
            e_d = (gx.cpu().double() - gref).abs().max().item() / gref.abs().max().item()
        err[m] = (e_f, e_d)
    assert err["x3"][0] <= max(4 * err["f32"][0], 1e-6) and err["x3"][0] <= 5e-6, err
    if err["x3"][1] is not None:
        assert err["x3"][1] <= max(4 * err["f32"][1], 1e-6) and err["x3"][1] <= 5e-6, err


def test_conv_x3_block_fwd_bwd_and_epilogues(x3_everywhere):
    """conv_block autograd (fwd + dgrad through conv_x3, wgrad on the fp32 kernel) and the DenseNet-style epilogues
    (channel-slice in/out, residual, alpha, accumulate) on the x3 path."""
    from irr_amd import conv as C
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 80, 32, 48, generator=g)
    w = torch.randn(64, 80, 3, 3, generator=g) * 0.05
    b = torch.randn(64, generator=g) * 0.1
    xc, wc, bc = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = _ref(xc, wc, bc, 3, 1, 1, True)
    go = torch.randn(yr.shape, generator=g)
    yr.backward(go)
    xd, wd, bd = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    assert C.x3_code(2, 80, 32, 48, 64, 3, 1, 1) and C.x3_code(2, 64, 32, 48, 80, 3, 1, 1)
    y = C.conv_block(xd, wd, bd, 1, 1, True)
    y.backward(go.cuda())
    tol = dict(rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(y.detach().cpu().numpy(), yr.detach().numpy(), **tol)
    np.testing.assert_allclose(xd.grad.cpu().numpy(), xc.grad.numpy(), **tol)
    np.testing.assert_allclose(wd.grad.cpu().numpy(), wc.grad.numpy(), rtol=1e-3, atol=2e-4 * max(float(wc.grad.abs().max()), 1.0))
    # channel-slice input/output + residual + alpha, then accumulate
    buf = torch.randn(2, 120, 32, 48, generator=g)
    w2 = torch.randn(40, 80, 3, 3, generator=g) * 0.05
    b2 = torch.randn(40, generator=g) * 0.1
    res = torch.randn(2, 40, 32, 48, generator=g)
    ref = res + 0.1 * _ref(buf[:, 40:], w2, b2, 3, 1, 1, True)
    d = buf.cuda()
    C.conv_forward(d[:, 40:], w2.cuda(), b2.cuda(), 1, 1, True, out=d[:, :40], res=res.cuda(), alpha=0.1)
    np.testing.assert_allclose(d[:, :40].cpu().numpy(), ref.numpy(), **tol)
    np.testing.assert_allclose(d[:, 40:].cpu().numpy(), buf[:, 40:].numpy())
    base = torch.randn(2, 40, 32, 48, generator=g)
    acc = base.cuda()
    C.conv_forward(d[:, 40:], w2.cuda(), None, 1, 1, False, out=acc, accumulate=True)
    np.testing.assert_allclose(acc.cpu().numpy(), (base + F.conv2d(buf[:, 40:], w2, None, padding=1)).numpy(), **tol)


def test_conv_x3_level4_linearity_and_routing():
    """BASELINE-size property check on the x3 path (bs16 level-4 decoder conv): default routing picks conv_x3,
    conv(2x) == 2 conv(x) bit-for-bit (power-of-two scaling commutes with the exact split), and the result agrees
    with the fp32-MFMA kernel to fp32 accuracy."""
    from irr_amd import conv as C
    torch.manual_seed(0)
    x = torch.randn(16, 115, 96, 112, device="cuda")
    w = torch.randn(128, 115, 3, 3, device="cuda") * 0.03
    C.set_math("x3")
    assert C.x3_code(16, 115, 96, 112, 128, 3, 1, 1) != 0
    y1 = C.conv_forward(x, w, None, 1, 1, False)
    y2 = C.conv_forward(2 * x, w, None, 1, 1, False)
    assert torch.equal(2 * y1, y2)
    C.set_math("f32")
    yf = C.conv_forward(x, w, None, 1, 1, False)
    C.set_math(C.DEFAULT_MATH)
    assert (y1 - yf).abs().max().item() < 2e-6 * yf.abs().max().item() * 4


WX3_CASES = [  # (Cin, Cout, B, H, W): every block shape (MW=4/3/2 and the swapped 8x1), all three (KG, R) unit shapes, channel tails
    (115, 128, 2, 24, 32), (40, 128, 1, 16, 48), (371, 96, 1, 12, 56), (64, 128, 2, 8, 40), (35, 96, 1, 20, 64),
    (565, 128, 1, 8, 112), (467, 64, 1, 16, 48), (64, 64, 2, 12, 56), (531, 32, 1, 16, 48), (64, 9, 2, 12, 56), (300, 32, 1, 8, 64),
    (32, 32, 2, 24, 64), (32, 9, 1, 16, 96), (24, 32, 2, 20, 32),          # one (co, ci) tile: pixels split over eight wave groups
    (243, 128, 2, 24, 28), (64, 64, 1, 12, 44), (371, 96, 1, 10, 36), (531, 32, 1, 24, 28),   # W % 8 == 4: half-empty last group
    (11, 32, 2, 16, 64),                                                                    # OccUpsampleNetwork.init_conv
    # one or two tiles, W % 32 != 0: four pixel wave groups (<1,1,2,4,4>, <2,1,2,4,4> direct and with exchanged roles)
    (32, 32, 2, 24, 56), (32, 32, 1, 16, 112), (24, 20, 2, 20, 40), (64, 32, 2, 12, 56), (48, 32, 1, 16, 48), (32, 64, 2, 16, 112),
    (20, 64, 1, 12, 56), (64, 32, 1, 24, 28), (32, 48, 1, 17, 36),
]


@pytest.mark.parametrize("case", WX3_CASES, ids=[f"{c[0]}to{c[1]}_{c[3]}x{c[4]}" for c in WX3_CASES])
def test_conv_wgrad_x3_is_fp32_faithful(case, x3_everywhere):
    """conv_wgrad_x3 (weight + bias gradient, alpha, accumulate-into-gw) against an fp64 reference: error in the fp32 class."""
    from irr_amd import conv as C, hip
    cin, cout, B, H, W = case
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(B, cin, H, W, generator=g)
    gy = torch.randn(B, cout, H, W, generator=g)
    ref = torch.nn.grad.conv2d_weight(x.double(), (cout, cin, 3, 3), gy.double(), padding=1)
    bref = gy.double().sum(dim=(0, 2, 3))
    assert hip.lib().irr_conv2d_wgrad_x3_eligible(B, cin, H, W, cout, 3, 1, 1) != 0
    base = torch.randn(cout, cin, 3, 3, generator=g)
    gw = base.clone().cuda()
    gb = torch.zeros(cout, device="cuda")
    C.conv_wgrad(x.cuda(), gy.cuda(), (cout, cin, 3, 3), 1, 1, gw=gw, gbias=gb, alpha=0.5)
    e_w = ((gw.cpu() - base).double() * 2 - ref).abs().max().item() / ref.abs().max().item()
    e_b = (gb.cpu().double() * 2 - bref).abs().max().item() / bref.abs().max().item()
    assert e_w <= 3e-6 and e_b <= 3e-6, (e_w, e_b)


X3S_CASES = [(32, 32, 2, 24, 64), (32, 32, 1, 70, 92), (24, 32, 1, 16, 32), (32, 9, 2, 16, 96), (16, 16, 2, 24, 64), (16, 32, 1, 16, 96)]


@pytest.mark.parametrize("case", X3S_CASES, ids=[f"{c[0]}to{c[1]}_{c[3]}x{c[4]}" for c in X3S_CASES])
def test_conv_x3_streaming_kernel_epilogues(case, x3_everywhere):
    """conv_x3s_kernel (persistent producer/consumer kernel of the 32-channel layers): bias+LeakyReLU, residual+alpha,
    accumulate, and the data-gradient launch with residual + alpha + accumulate + LeakyReLU'-mask, against fp64."""
    from irr_amd import conv as C
    cin, cout, B, H, W = case
    g = torch.Generator().manual_seed(cin * 3 + cout)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * 0.1
    b = torch.randn(cout, generator=g)
    res = torch.randn(B, cout, H, W, generator=g)
    base = torch.randn(B, cout, H, W, generator=g)
    assert C.x3_code(B, cin, H, W, cout, 3, 1, 1) == 9001
    conv = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    tol = 2e-6 * float(conv.abs().max())
    y = C.conv_forward(x.cuda(), w.cuda(), b.cuda(), 1, 1, True)
    assert (y.cpu().double() - F.leaky_relu(conv, 0.1)).abs().max().item() <= tol
    y = C.conv_forward(x.cuda(), w.cuda(), b.cuda(), 1, 1, False, res=res.cuda(), alpha=0.1)
    assert (y.cpu().double() - (res.double() + 0.1 * conv)).abs().max().item() <= tol
    acc = base.clone().cuda()
    C.conv_forward(x.cuda(), w.cuda(), None, 1, 1, False, out=acc, accumulate=True)
    assert (acc.cpu().double() - (base.double() + F.conv2d(x.double(), w.double(), None, padding=1))).abs().max().item() <= tol
    if C.x3_code(B, cout, H, W, cin, 3, 1, 1) == 9001:
        gy = torch.randn(B, cout, H, W, generator=g)
        g0 = torch.randn(B, cin, H, W, generator=g)
        r2 = torch.randn(B, cin, H, W, generator=g)
        m2 = torch.randn(B, cin, H, W, generator=g)
        gx = g0.clone().cuda()
        C.conv_dgrad(gy.cuda(), w.cuda(), 1, 1, (H, W), gx=gx, accumulate=True, res=r2.cuda(), alpha=0.1, mask=m2.cuda(), nmask=cin)
        ref = (g0.double() + r2.double() + 0.1 * torch.nn.grad.conv2d_input(x.shape, w.double(), gy.double(), padding=1)) * \
            torch.where(m2 > 0, 1.0, 0.1).double()
        assert (gx.cpu().double() - ref).abs().max().item() <= 2e-6 * float(ref.abs().max())


X3K_CASES = [(565, 128, 1, 64, 12, 14), (128, 128, 2, 64, 12, 14), (467, 64, 1, 64, 24, 28), (115, 128, 1, 64, 12, 14)]


@pytest.mark.parametrize("case", X3K_CASES, ids=[f"{c[0]}to{c[1]}d{c[2]}_{c[4]}x{c[5]}" for c in X3K_CASES])
def test_conv_x3_k_split_small_levels(case):
    """Small pyramid levels at the BASELINE batch (default routing): conv_x3_kernel with blockIdx.z splitting the channel
    chunks + x3_splitk_epilogue_kernel.  Forward with bias + LeakyReLU, residual + alpha, accumulate, and the data gradient
    with accumulate + LeakyReLU'-mask, against fp64."""
    from irr_amd import conv as C, hip
    cin, cout, dil, B, H, W = case
    C.x3_code(1, 64, 8, 8, 64, 3, 1, 1)                       # (applies IRR_X3_MIN_BLOCKS once, if set)
    old = hip.lib().irr_conv_x3_set_min_blocks(384)          # the K split belongs to the default routing
    C.set_math("x3")
    try:
        _k_split_checks(C, hip, cin, cout, dil, B, H, W)
    finally:
        hip.lib().irr_conv_x3_set_min_blocks(old)
        C.set_math(C.DEFAULT_MATH)


def _k_split_checks(C, hip, cin, cout, dil, B, H, W):
    assert hip.lib().irr_conv2d_fwd_x3_ws_elems(B, cin, H, W, cout, dil) > 0, "case must take the K-split path"
    assert C.x3_code(B, cin, H, W, cout, 3, 1, dil) != 0
    g = torch.Generator().manual_seed(cin + 3 * cout)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    res = torch.randn(B, cout, H, W, generator=g)
    base = torch.randn(B, cout, H, W, generator=g)
    conv = F.conv2d(x.double(), w.double(), b.double(), padding=dil, dilation=dil)
    tol = 3e-6 * float(conv.abs().max())
    y = C.conv_forward(x.cuda(), w.cuda(), b.cuda(), 1, dil, True)
    assert (y.cpu().double() - F.leaky_relu(conv, 0.1)).abs().max().item() <= tol
    y = C.conv_forward(x.cuda(), w.cuda(), b.cuda(), 1, dil, False, res=res.cuda(), alpha=0.1)
    assert (y.cpu().double() - (res.double() + 0.1 * conv)).abs().max().item() <= tol
    acc = base.clone().cuda()
    C.conv_forward(x.cuda(), w.cuda(), None, 1, dil, False, out=acc, accumulate=True)
    ref = base.double() + F.conv2d(x.double(), w.double(), None, padding=dil, dilation=dil)
    assert (acc.cpu().double() - ref).abs().max().item() <= tol
    if hip.lib().irr_conv2d_fwd_x3_ws_elems(B, cout, H, W, cin, dil) > 0:
        gy = torch.randn(B, cout, H, W, generator=g)
        g0 = torch.randn(B, cin, H, W, generator=g)
        m2 = torch.randn(B, cin, H, W, generator=g)
        gx = g0.clone().cuda()
        C.conv_dgrad(gy.cuda(), w.cuda(), 1, dil, (H, W), gx=gx, accumulate=True, mask=m2.cuda(), nmask=cin)
        gref = (g0.double() + torch.nn.grad.conv2d_input(x.shape, w.double(), gy.double(), padding=dil, dilation=dil)) * \
            torch.where(m2 > 0, 1.0, 0.1).double()
        assert (gx.cpu().double() - gref).abs().max().item() <= 3e-6 * float(gref.abs().max())


@pytest.mark.parametrize("cout,dil", [(2, 1), (1, 1), (2, 2), (1, 4)])
def test_conv_smallco_dgrad_accumulate_mask(cout, dil):
    """irr_conv2d_smallco_dgrad_f32 (the conv_last / context-tail data gradient): plain, and accumulate + LeakyReLU'-mask
    into a channel slice of a larger gradient buffer (the DenseNet use, conv._DenseEstimatorFn.backward)."""
    from irr_amd import conv as C
    g = torch.Generator().manual_seed(40 + cout + dil)
    B, cin, H, W = 2, 37, 19, 23
    w = torch.randn(cout, cin, 3, 3, generator=g) * 0.2
    gy = torch.randn(B, cout, H, W, generator=g)
    ref = torch.nn.grad.conv2d_input((B, cin, H, W), w.double(), gy.double(), padding=dil, dilation=dil)
    gx = C.conv_dgrad(gy.cuda(), w.cuda(), 1, dil, (H, W))
    np.testing.assert_allclose(gx.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)
    big = torch.randn(B, cin + 5, H, W, generator=g)
    msk = torch.randn(B, cin + 5, H, W, generator=g)
    G = big.clone().cuda()
    C.conv_dgrad(gy.cuda(), w.cuda(), 1, dil, (H, W), gx=G[:, :cin], accumulate=True, mask=msk.cuda()[:, :cin], nmask=11)
    want = big.double().clone()
    want[:, :cin] += ref
    want[:, :11] *= torch.where(msk[:, :11] > 0, 1.0, 0.1).double()
    np.testing.assert_allclose(G.cpu().numpy(), want.numpy(), rtol=1e-5, atol=1e-5)


WX3_DIL_CASES = [(128, 128, 1, 24, 32, 2), (128, 128, 2, 19, 48, 4), (128, 96, 1, 40, 56, 8), (96, 64, 1, 50, 48, 16),
                 (96, 64, 2, 48, 56, 16), (96, 64, 1, 35, 40, 16),     # dilation 16 on rows of an odd group count: the (1, 2) unit (round 4)
                 (128, 128, 3, 24, 28, 2), (128, 128, 2, 24, 28, 4), (128, 96, 3, 24, 28, 8), (96, 64, 2, 24, 28, 16), (128, 128, 1, 17, 36, 4),   # W % 8 == 4
                 (128, 128, 1, 20, 56, 2), (128, 128, 1, 33, 64, 4), (128, 96, 2, 24, 64, 8), (96, 64, 2, 40, 64, 16)]


@pytest.mark.parametrize("case", WX3_DIL_CASES, ids=[f"{c[0]}to{c[1]}d{c[5]}_{c[3]}x{c[4]}" for c in WX3_DIL_CASES])
def test_conv_wgrad_x3_dilated(case, x3_everywhere):
    """The dilated context-network layers on conv_wgrad_x3_kernel<..., DIL> (row-residue walk, dword / group shifted taps)."""
    from irr_amd import conv as C, hip
    cin, cout, B, H, W, dil = case
    g = torch.Generator().manual_seed(cin + cout + dil)
    x = torch.randn(B, cin, H, W, generator=g)
    gy = torch.randn(B, cout, H, W, generator=g)
    ref = torch.nn.grad.conv2d_weight(x.double(), (cout, cin, 3, 3), gy.double(), padding=dil, dilation=dil)
    bref = gy.double().sum(dim=(0, 2, 3))
    assert hip.lib().irr_conv2d_wgrad_x3_eligible(B, cin, H, W, cout, 3, 1, dil) == 5000 + dil
    gw = torch.zeros(cout, cin, 3, 3, device="cuda")
    gb = torch.zeros(cout, device="cuda")
    C.conv_wgrad(x.cuda(), gy.cuda(), (cout, cin, 3, 3), 1, dil, gw=gw, gbias=gb)
    assert (gw.cpu().double() - ref).abs().max().item() <= 3e-6 * ref.abs().max().item()
    assert (gb.cpu().double() - bref).abs().max().item() <= 3e-6 * bref.abs().max().item()


@pytest.mark.parametrize("cout", [1, 2])
def test_conv_smallco_quad_kernels(cout):
    """4-pixel small-Cout kernels (W % 4 == 0): forward with bias / residual / alpha / accumulate on channel-slice views, and
    the weight + bias gradient, against torch on the host."""
    from irr_amd import conv as C
    g = torch.Generator().manual_seed(70 + cout)
    B, cin, H, W = 2, 45, 14, 36
    big = torch.randn(B, cin + 7, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * 0.2
    b = torch.randn(cout, generator=g)
    res = torch.randn(B, cout, H, W, generator=g)
    x = big[:, 7:]
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    xd = big.cuda()[:, 7:]
    y = C.conv_forward(xd, w.cuda(), b.cuda(), 1, 1, False, res=res.cuda(), alpha=0.5)
    np.testing.assert_allclose(y.cpu().numpy(), (res.double() + 0.5 * ref).numpy(), rtol=1e-5, atol=1e-5)
    y = C.conv_forward(xd, w.cuda(), b.cuda(), 1, 1, True)
    np.testing.assert_allclose(y.cpu().numpy(), F.leaky_relu(ref, 0.1).numpy(), rtol=1e-5, atol=1e-5)
    base = torch.randn(B, cout, H, W, generator=g)
    acc = base.clone().cuda()
    C.conv_forward(xd, w.cuda(), None, 1, 1, False, out=acc, accumulate=True)
    np.testing.assert_allclose(acc.cpu().numpy(), (base.double() + F.conv2d(x.double(), w.double(), None, padding=1)).numpy(), rtol=1e-5, atol=1e-5)
    gy = torch.randn(B, cout, H, W, generator=g)
    gw = torch.zeros(cout, cin, 3, 3, device="cuda")
    gb = torch.zeros(cout, device="cuda")
    C.conv_wgrad(xd, gy.cuda(), (cout, cin, 3, 3), 1, 1, gw=gw, gbias=gb, alpha=2.0)
    wref = torch.nn.grad.conv2d_weight(x.double(), (cout, cin, 3, 3), gy.double(), padding=1)
    np.testing.assert_allclose(gw.cpu().numpy(), 2 * wref.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(gb.cpu().numpy(), 2 * gy.double().sum(dim=(0, 2, 3)).numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("shape", [(64, 6, 7), (64, 12, 14), (3, 9, 11)])
@pytest.mark.parametrize("cout", [1, 2])
def test_conv_smallco_tiny_planes_channel_slices(cout, shape):
    """conv_last at the 6x7 / 12x14 pyramid levels (W % 4 != 0, 565 input channels, BASELINE batch): the scalar kernel with 16
    channel slices per 64 pixels (conv_smallco_fwd_sl_kernel) -- bias + LeakyReLU, residual + alpha and accumulate on a
    channel-slice view, against fp64 on the host."""
    from irr_amd import conv as C
    B, H, W = shape
    cin = 563 + cout % 2 * 2                                  # 565 -> 1 and 563 -> 2
    g = torch.Generator().manual_seed(300 + cout + H)
    big = torch.randn(B, cin + 3, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
    b = torch.randn(cout, generator=g)
    res = torch.randn(B, cout, H, W, generator=g)
    x = big[:, 3:]
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    xd = big.cuda()[:, 3:]
    y = C.conv_forward(xd, w.cuda(), b.cuda(), 1, 1, True)
    np.testing.assert_allclose(y.cpu().numpy(), F.leaky_relu(ref, 0.1).numpy(), rtol=1e-5, atol=2e-5)
    y = C.conv_forward(xd, w.cuda(), b.cuda(), 1, 1, False, res=res.cuda(), alpha=0.5)
    np.testing.assert_allclose(y.cpu().numpy(), (res.double() + 0.5 * ref).numpy(), rtol=1e-5, atol=2e-5)
    base = torch.randn(B, cout, H, W, generator=g)
    acc = base.clone().cuda()
    C.conv_forward(xd, w.cuda(), None, 1, 1, False, out=acc, accumulate=True)
    np.testing.assert_allclose(acc.cpu().numpy(), (base.double() + F.conv2d(x.double(), w.double(), None, padding=1)).numpy(),
                               rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("cout", [1, 2])
def test_conv_smallco_heads_large_level(cout):
    """The head kernels in their large-level configuration (>= 400000 pixels: four output rows per thread, channel split
    over 4 waves; height divisible by 4 but not by 8, ragged last row group excluded by H % 4 == 0): forward with bias,
    weight + bias gradient and the accumulate + mask data gradient against torch's fp32 GPU convolution / fp64 on a slice."""
    from irr_amd import conv as C
    g = torch.Generator().manual_seed(90 + cout)
    B, cin, H, W = 16, 70, 100, 256                          # 409600 pixels
    x = torch.randn(B, cin, H, W, generator=g).cuda()
    w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.1).cuda()
    b = torch.randn(cout, generator=g).cuda()
    y = C.conv_forward(x, w, b, 1, 1, False)
    ref = F.conv2d(x[:2].double().cpu(), w.double().cpu(), b.double().cpu(), padding=1)
    np.testing.assert_allclose(y[:2].cpu().numpy(), ref.numpy(), rtol=1e-5, atol=2e-5)
    ref_last = F.conv2d(x[-1:].double().cpu(), w.double().cpu(), b.double().cpu(), padding=1)
    np.testing.assert_allclose(y[-1:].cpu().numpy(), ref_last.numpy(), rtol=1e-5, atol=2e-5)
    gy = torch.randn(B, cout, H, W, generator=g).cuda()
    gw = torch.zeros(cout, cin, 3, 3, device="cuda")
    gb = torch.zeros(cout, device="cuda")
    C.conv_wgrad(x, gy, (cout, cin, 3, 3), 1, 1, gw=gw, gbias=gb)
    wref = torch.nn.grad.conv2d_weight(x.double().cpu(), (cout, cin, 3, 3), gy.double().cpu(), padding=1)
    np.testing.assert_allclose(gw.cpu().numpy(), wref.numpy(), rtol=2e-4, atol=2e-4 * float(wref.abs().max()))
    np.testing.assert_allclose(gb.cpu().numpy(), gy.double().sum(dim=(0, 2, 3)).cpu().numpy(), rtol=2e-4, atol=1e-2)
    g0 = torch.randn(B, cin, H, W, generator=g).cuda()
    gx = g0.clone()
    C.conv_dgrad(gy, w, 1, 1, (H, W), gx=gx, accumulate=True, mask=x, nmask=32)
    gref = torch.nn.grad.conv2d_input(x[:2].shape, w.double().cpu(), gy[:2].double().cpu(), padding=1) + g0[:2].double().cpu()
    gref[:, :32] *= torch.where(x[:2, :32].cpu() > 0, 1.0, 0.1).double()
    np.testing.assert_allclose(gx[:2].cpu().numpy(), gref.numpy(), rtol=1e-5, atol=2e-5)


def test_x3_family_baseline_size_properties():
    """Size-independent properties at BASELINE configs[2] layer sizes (bs32 -> 2B = 64 samples), where a host reference
    is too slow: (i) weight gradient of a level-4 decoder layer: linear in gy and additive over the batch; (ii) the
    streaming 32-channel kernel at full resolution: conv(2x) == 2 conv(x) bit for bit, and equal to the fp32-MFMA kernel
    to fp32 accuracy."""
    from irr_amd import conv as C, hip
    C.set_math("x3")
    torch.manual_seed(1)
    B, cin, cout, H, W = 64, 128, 128, 96, 112
    x = torch.randn(B, cin, H, W, device="cuda")
    gy = torch.randn(B, cout, H, W, device="cuda")
    assert hip.lib().irr_conv2d_wgrad_x3_eligible(B, cin, H, W, cout, 3, 1, 1) != 0

    def wg(xx, gg):
        gw = torch.zeros(cout, cin, 3, 3, device="cuda")
        C.conv_wgrad(xx, gg, (cout, cin, 3, 3), 1, 1, gw=gw)
        return gw
    full = wg(x, gy)
    scale = full.abs().max().item()
    assert (wg(x, 2 * gy) - 2 * full).abs().max().item() <= 2e-6 * scale * 2
    halves = wg(x[:32], gy[:32]) + wg(x[32:], gy[32:])
    assert (halves - full).abs().max().item() <= 4e-6 * scale
    del x, gy
    x = torch.randn(16, 32, 384, 448, device="cuda")
    w = torch.randn(32, 32, 3, 3, device="cuda") * 0.05
    assert C.x3_code(16, 32, 384, 448, 32, 3, 1, 1) == 9001
    y1 = C.conv_forward(x, w, None, 1, 1, False)
    y2 = C.conv_forward(2 * x, w, None, 1, 1, False)
    assert torch.equal(2 * y1, y2)
    C.set_math("f32")
    yf = C.conv_forward(x, w, None, 1, 1, False)
    C.set_math(C.DEFAULT_MATH)
    assert (y1 - yf).abs().max().item() <= 4e-6 * yf.abs().max().item()


BASELINE_X3 = [  # (Cin, Cout, dil, B, H, W): BASELINE configs[2] launches of the configurations chosen at full size only
    (128, 96, 8, 64, 96, 112),      # row-folded dilation 8, three co-tiles
    (96, 128, 8, 64, 96, 112),      # its data gradient shape
    (64, 96, 16, 64, 96, 112),      # dilation 16 data gradient shape
    (128, 128, 4, 64, 48, 56),      # row-folded dilation 4 at level 3
    (243, 128, 1, 64, 96, 112),     # 224-pixel tiles chosen by the round-aware rule
    (16, 16, 1, 64, 192, 224),      # pyramid 16 -> 16 on the streaming kernel (two chunks, second with zero weights)
    (115, 128, 1, 64, 12, 14),      # K split over blockIdx.z
]


@pytest.mark.parametrize("case", BASELINE_X3, ids=[f"{c[0]}to{c[1]}d{c[2]}_{c[4]}x{c[5]}" for c in BASELINE_X3])
def test_x3_matches_fp32_kernel_at_baseline_sizes(case):
    """Default routing at the BASELINE batch: the x3 launch (forward with bias + LeakyReLU, and the transposed-weight data
    gradient) against the fp32-MFMA kernel on the same operands -- both are within 1-2e-6 of fp64 on small shapes, so they
    must agree to 4e-6 of the output range here, where a host reference is too slow."""
    from irr_amd import conv as C, hip
    cin, cout, dil, B, H, W = case
    C.x3_code(1, 64, 8, 8, 64, 3, 1, 1)
    old = hip.lib().irr_conv_x3_set_min_blocks(384)
    try:
        torch.manual_seed(cin + cout + dil)
        x = torch.randn(B, cin, H, W, device="cuda")
        w = torch.randn(cout, cin, 3, 3, device="cuda") * (2.0 / (cin * 9)) ** 0.5
        b = torch.randn(cout, device="cuda") * 0.1
        gy = torch.randn(B, cout, H, W, device="cuda")
        C.set_math("x3")
        assert C.x3_code(B, cin, H, W, cout, 3, 1, dil) != 0
        y3 = C.conv_forward(x, w, b, 1, dil, True)
        g3 = C.conv_dgrad(gy, w, 1, dil, (H, W)) if C.x3_code(B, cout, H, W, cin, 3, 1, dil) else None
        C.set_math("f32")
        yf = C.conv_forward(x, w, b, 1, dil, True)
        gf = C.conv_dgrad(gy, w, 1, dil, (H, W))
    finally:
        hip.lib().irr_conv_x3_set_min_blocks(old)
        C.set_math(C.DEFAULT_MATH)
    assert (y3 - yf).abs().max().item() <= 4e-6 * yf.abs().max().item()
    if g3 is not None:
        assert (g3 - gf).abs().max().item() <= 4e-6 * gf.abs().max().item()


BASELINE_WX3 = [  # (Cin, Cout, dil, B, H, W)
    (64, 32, 1, 64, 96, 112), (32, 32, 1, 64, 96, 112), (32, 64, 1, 64, 96, 112), (565, 128, 1, 64, 48, 56), (128, 96, 8, 64, 96, 112),
    (115, 128, 1, 64, 96, 112), (32, 32, 1, 64, 192, 224), (531, 32, 1, 64, 48, 56),
]


@pytest.mark.parametrize("case", BASELINE_WX3, ids=[f"{c[0]}to{c[1]}d{c[2]}_{c[4]}x{c[5]}" for c in BASELINE_WX3])
def test_wgrad_x3_matches_fp32_kernel_at_baseline_sizes(case):
    """Weight + bias gradient at the BASELINE batch: x3 kernels (partial images + fixed-order reduce; K-split and role-swapped
    variants, balanced column chunks, dilated walks) against the fp32-MFMA kernels on the same operands."""
    from irr_amd import conv as C, hip
    cin, cout, dil, B, H, W = case
    torch.manual_seed(cin * 3 + cout + dil)
    x = torch.randn(B, cin, H, W, device="cuda")
    gy = torch.randn(B, cout, H, W, device="cuda")
    out = {}
    try:
        for m in ("x3", "f32"):
            C.set_math(m)
            if m == "x3":
                assert hip.lib().irr_conv2d_wgrad_x3_eligible(B, cin, H, W, cout, 3, 1, dil) != 0
            gw = torch.zeros(cout, cin, 3, 3, device="cuda")
            gb = torch.zeros(cout, device="cuda")
            C.conv_wgrad(x, gy, (cout, cin, 3, 3), 1, dil, gw=gw, gbias=gb)
            out[m] = (gw, gb)
    finally:
        C.set_math(C.DEFAULT_MATH)
    sw, sb = out["f32"][0].abs().max().item(), out["f32"][1].abs().max().item()
    assert (out["x3"][0] - out["f32"][0]).abs().max().item() <= 4e-6 * sw
    assert (out["x3"][1] - out["f32"][1]).abs().max().item() <= 1e-5 * sb


@pytest.mark.parametrize("shape", [(2, 32, 40, 64), (1, 24, 22, 96), (3, 32, 8, 32)])
def test_conv_x3s_second_output_is_the_unfused_pair(shape, x3_everywhere):
    """irr_conv2d_fwd_x3_dual (conv_forward_skip): e = lrelu(conv(x) + b) and y = skip + e from one launch of the streaming kernel
    are bit-identical to the plain launch followed by torch.add (the pair it replaces in the occlusion upsampler), ragged tiles and
    a channel-slice view as input included; a problem the streaming kernel does not take falls back to that pair."""
    from irr_amd import conv as C
    B, cin, H, W = shape
    g = torch.Generator().manual_seed(H * W + cin)
    big = torch.randn(B, cin + 5, H, W, generator=g).cuda()
    x = big[:, 5:]
    w = (torch.randn(32, cin, 3, 3, generator=g) * 0.1).cuda()
    b = torch.randn(32, generator=g).cuda()
    skip = torch.randn(B, 32, H, W, generator=g).cuda()
    assert C.x3_code(B, cin, H, W, 32, 3, 1, 1) == 9001
    e, y = C.conv_forward_skip(x, w, b, True, skip)
    e_ref = C.conv_forward(x, w, b, 1, 1, True)
    assert torch.equal(e, e_ref)
    assert torch.equal(y, torch.add(skip, e_ref))
    ref = F.leaky_relu(F.conv2d(x.double().cpu(), w.double().cpu(), b.double().cpu(), padding=1), 0.1)
    np.testing.assert_allclose(e.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)
    # not a streaming-kernel problem (64 input channels): the fallback pair
    x64 = torch.randn(B, 64, H, W, generator=g).cuda()
    w64 = (torch.randn(32, 64, 3, 3, generator=g) * 0.1).cuda()
    e2, y2 = C.conv_forward_skip(x64, w64, b, True, skip)
    assert torch.equal(y2, skip + e2) and torch.equal(e2, C.conv_forward(x64, w64, b, 1, 1, True))


@pytest.mark.parametrize("nmask", [0, 40, 64])
def test_conv_x3s_two_cotiles(nmask, x3_everywhere):
    """conv_x3s_kernel launched once per 32-channel co-tile (Cout = 64, Cin = 32): the forward epilogues and the data gradient
    of a 64 -> 32 layer with residual + alpha + accumulate and a LeakyReLU'-mask that ends inside the second co-tile."""
    from irr_amd import conv as C
    g = torch.Generator().manual_seed(123 + nmask)
    B, H, W = 2, 24, 64
    x = torch.randn(B, 32, H, W, generator=g)
    w = torch.randn(64, 32, 3, 3, generator=g) * 0.1
    b = torch.randn(64, generator=g)
    res = torch.randn(B, 64, H, W, generator=g)
    assert C.x3_code(B, 32, H, W, 64, 3, 1, 1) == 9001
    conv = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    tol = 2e-6 * float(conv.abs().max())
    y = C.conv_forward(x.cuda(), w.cuda(), b.cuda(), 1, 1, True)
    assert (y.cpu().double() - F.leaky_relu(conv, 0.1)).abs().max().item() <= tol
    y = C.conv_forward(x.cuda(), w.cuda(), b.cuda(), 1, 1, False, res=res.cuda(), alpha=0.1)
    assert (y.cpu().double() - (res.double() + 0.1 * conv)).abs().max().item() <= tol
    # data gradient of a 64 -> 32 layer: gy has 32 channels, gx 64
    wl = torch.randn(32, 64, 3, 3, generator=g) * 0.1
    gy = torch.randn(B, 32, H, W, generator=g)
    g0 = torch.randn(B, 64, H, W, generator=g)
    r2 = torch.randn(B, 64, H, W, generator=g)
    m2 = torch.randn(B, 64, H, W, generator=g)
    gx = g0.clone().cuda()
    C.conv_dgrad(gy.cuda(), wl.cuda(), 1, 1, (H, W), gx=gx, accumulate=True, res=r2.cuda(), alpha=0.1,
                 mask=m2.cuda() if nmask else None, nmask=nmask)
    ref = g0.double() + r2.double() + 0.1 * torch.nn.grad.conv2d_input((B, 64, H, W), wl.double(), gy.double(), padding=1)
    if nmask:
        ref[:, :nmask] *= torch.where(m2[:, :nmask] > 0, 1.0, 0.1).double()
    assert (gx.cpu().double() - ref).abs().max().item() <= 2e-6 * float(ref.abs().max())


# ---- the DenseNet estimators on the x3 family: combined-column data gradient (irr_conv_pack_weights_x3_sub) -----------
def _dense_ref64(x, base, ws, bs):
    """models/pwc_modules.py:153-170 / 190-207 in fp64 torch: (cat([x5, est]), est), est = base + conv_last(x5)"""
    cur = x
    for i in range(5):
        cur = torch.cat([F.leaky_relu(F.conv2d(cur, ws[i], bs[i], padding=1), 0.1), cur], dim=1)
    est = base + F.conv2d(cur, ws[5], bs[5], padding=1)
    return torch.cat([cur, est], dim=1), est


@pytest.mark.parametrize("which,B,H,W", [("flow_estimators", 2, 32, 48), ("occ_estimators", 1, 40, 56), ("flow_estimators", 3, 24, 28)])
def test_dense_estimator_on_x3_vs_fp64(which, B, H, W, x3_everywhere):
    """FlowEstimatorDense / OccEstimatorDense ``forward_residual`` (the model's fast path) forward AND backward with every
    layer on the x3 family -- five forward launches into channel slices, conv_last head, and the column-wise backward:
    pack mode 2 (``irr_conv_pack_weights_x3_sub``) -> conv_x3 with accumulate + LeakyReLU'-mask epilogue on slices of the
    gradient buffer, conv_wgrad_x3 for dW / db -- against an fp64 torch restatement of the reference module."""
    import irr_amd
    from irr_amd import conv as C
    import types
    torch.manual_seed(0)
    m = irr_amd.PWCNet(types.SimpleNamespace(batch_size=B, model_div_flow=0.05)).cuda().train()
    est = getattr(m, which)
    E = 2 if which == "flow_estimators" else 1
    cin0 = 81 + 32 + E
    g = torch.Generator().manual_seed(B * 100 + H)
    for layer in (est.conv1, est.conv2, est.conv3, est.conv4, est.conv5, est.conv_last):     # MSRA biases are zero: randomise them
        with torch.no_grad():
            layer.bias.copy_(torch.randn(layer.bias.shape, generator=g) * 0.1)
    x = torch.randn(B, cin0, H, W, generator=g)
    base = torch.randn(B, E, H, W, generator=g)
    gi = torch.randn(B, 448 + cin0 + E, H, W, generator=g)
    gf = torch.randn(B, E, H, W, generator=g)
    layers = (est.conv1, est.conv2, est.conv3, est.conv4, est.conv5, est.conv_last)
    ws = [l.weight.detach().cpu().double().requires_grad_(True) for l in layers]
    bs = [l.bias.detach().cpu().double().requires_grad_(True) for l in layers]
    x64, b64 = x.double().requires_grad_(True), base.double().requires_grad_(True)
    buf_r, est_r = _dense_ref64(x64, b64, ws, bs)
    ((buf_r * gi.double()).sum() + (est_r * gf.double()).sum()).backward()
    C.LAUNCHES.clear()
    xd, bd = x.cuda().requires_grad_(True), base.cuda().requires_grad_(True)
    buf, out = est.forward_residual(xd, bd)
    ((buf * gi.cuda()).sum() + (out * gf.cuda()).sum()).backward()
    assert C.LAUNCHES["dense_column_x3"] == 5 and C.LAUNCHES["dense_column_f32"] == 0, dict(C.LAUNCHES)
    assert C.LAUNCHES["fwd_x3"] + C.LAUNCHES["fwd_x3s"] >= 5 and C.LAUNCHES["fwd_f32"] == 0, dict(C.LAUNCHES)
    if W % 4 == 0 and W >= 24:
        assert C.LAUNCHES["wgrad_x3"] == 5 and C.LAUNCHES["wgrad_f32"] == 0, dict(C.LAUNCHES)

    def close(a, r, tol, what):
        e = (a.detach().cpu().double() - r.detach()).abs().max().item() / max(r.detach().abs().max().item(), 1e-30)
        assert e <= tol, (what, e)

    close(buf, buf_r, 1e-5, "buf")
    close(out, est_r, 1e-5, "est")
    close(xd.grad, x64.grad, 2e-5, "gx")
    close(bd.grad, b64.grad, 2e-5, "gbase")
    for i, l in enumerate(layers):
        close(l.weight.grad, ws[i].grad, 2e-5, f"dW{i}")
        close(l.bias.grad, bs[i].grad, 2e-5, f"db{i}")


def test_x3_sub_pack_combined_column_launch_vs_fp64(x3_everywhere):
    """ONE combined-column launch in isolation: rows of the packed matrix come from several layers' transposed + flipped
    weights (pack mode 2), the launch accumulates into its target slice and applies LeakyReLU'(mask) -- the dominant
    kernel of the BASELINE step (conv_x3_kernel<4,1,7,352> on the c1 / x columns)."""
    from irr_amd import conv as C, hip
    g = torch.Generator().manual_seed(17)
    B, H, W, cin0 = 2, 32, 48, 115
    ctot = 448 + cin0
    grow = (128, 128, 96, 64, 32)
    in0 = [448, 320, 192, 96, 32]
    row0 = {5: 0, 4: 32, 3: 96, 2: 192, 1: 320}
    ws = [torch.randn(grow[i], ctot - in0[i], 3, 3, generator=g) * (2.0 / (9 * (ctot - in0[i]))) ** 0.5 for i in range(5)]
    G0 = torch.randn(B, ctot, H, W, generator=g)
    act = torch.randn(B, ctot, H, W, generator=g)
    bounds = [(32, 96), (96, 192), (192, 320), (320, 448), (448, ctot)]
    packs = C._dense_column_packs([w.cuda() for w in ws], cin0, (True,) * 5)
    for k_, (t0, t1) in enumerate(bounds):
        n = t1 - t0
        last = k_ == 4
        ref = G0[:, t0:t1].double().clone()
        for i in (5, 4, 3, 2, 1):
            if in0[i - 1] > t0:
                continue
            w = ws[i - 1].double()
            gy = G0[:, row0[i]:row0[i] + w.shape[0]].double()
            full = torch.nn.grad.conv2d_input((B, w.shape[1], H, W), w, gy, padding=1)
            ref += full[:, t0 - in0[i - 1]:t1 - in0[i - 1]]
        if not last:
            ref *= torch.where(act[:, t0:t1] > 0, 1.0, 0.1).double()
        Gd = G0.clone().cuda()
        ad = act.cuda()
        assert C.x3_code(B, t0, H, W, n, 3, 1, 1) != 0
        margs = (None, 0, 0) if last else (hip.ptr(ad[:, t0:t1]), hip.bs(ad), n)
        C._call_conv(("irr_conv2d_fwd_x3", hip.ptr(Gd), hip.ptr(packs[k_]), None, None, hip.ptr(Gd[:, t0:t1]), B, t0, H, W, n, 1,
                      hip.bs(Gd), hip.bs(Gd), 0, 0, 1.0, 1, *margs, hip.stream()))
        e = (Gd[:, t0:t1].cpu().double() - ref).abs().max().item() / ref.abs().max().item()
        assert e <= 5e-6, (k_, e)
        assert torch.equal(Gd[:, :t0].cpu(), G0[:, :t0]) and torch.equal(Gd[:, t1:].cpu(), G0[:, t1:])      # nothing else touched


def test_deferred_batched_fold_is_bit_identical_to_immediate():
    """The fold of the weight-gradient partial images, deferred into ONE batched launch (conv.ReduceBatch, what the
    asynchronous lane uses), against the launch-by-launch fold: same bits, for the x3 family (plain, role-swapped,
    dilated, K-split) and the fp32 family; a gradient that already has a pending fold forces a flush."""
    from irr_amd import conv as C, hip
    old = hip.lib().irr_conv_x3_set_min_blocks(0)
    try:
        cases = [(64, 128, 2, 24, 32, 1), (40, 32, 1, 16, 48, 1), (32, 32, 2, 24, 64, 1), (128, 96, 1, 40, 56, 8), (24, 48, 2, 12, 14, 1),
                 (96, 64, 1, 20, 28, 1)]
        g = torch.Generator().manual_seed(3)
        data = []
        for cin, cout, B, H, W, dil in cases:
            data.append((torch.randn(B, cin, H, W, generator=g).cuda(), torch.randn(B, cout, H, W, generator=g).cuda(), cin, cout, dil))
        ref = []
        for x, gy, cin, cout, dil in data:
            gw, gb = torch.zeros(cout, cin, 3, 3, device="cuda"), torch.zeros(cout, device="cuda")
            C.conv_wgrad(x, gy, (cout, cin, 3, 3), 1, dil, gw=gw, gbias=gb, alpha=0.5)
            ref.append((gw, gb))
        batch = C.ReduceBatch()
        got = []
        for x, gy, cin, cout, dil in data:
            gw, gb = torch.zeros(cout, cin, 3, 3, device="cuda"), torch.zeros(cout, device="cuda")
            C.conv_wgrad(x, gy, (cout, cin, 3, 3), 1, dil, gw=gw, gbias=gb, alpha=0.5, defer=batch)
            got.append((gw, gb))
        assert batch.n == len(cases) and all(float(gw.abs().max()) == 0.0 for gw, _ in got)     # nothing folded yet
        assert batch.full_for(got[0][0]) and not batch.full_for(torch.zeros(4, device="cuda"))
        keep = batch.run()
        assert batch.n == 0 and len(keep) == 2 * len(cases)
        torch.cuda.synchronize()
        for (gw, gb), (rw, rb) in zip(got, ref):
            assert torch.equal(gw, rw) and float(rw.abs().max()) > 0
            np.testing.assert_allclose(gb.cpu().numpy(), rb.cpu().numpy(), rtol=1e-5, atol=1e-5)     # (bias sums use atomics)
        # two folds into ONE gradient in a single batch are refused by the library
        x, gy, cin, cout, dil = data[0]
        gw = torch.zeros(cout, cin, 3, 3, device="cuda")
        C.conv_wgrad(x, gy, (cout, cin, 3, 3), 1, dil, gw=gw, defer=batch)
        C.conv_wgrad(x, gy, (cout, cin, 3, 3), 1, dil, gw=gw, defer=batch)
        with pytest.raises(hip.HipError):
            batch.run()
    finally:
        hip.lib().irr_conv_x3_set_min_blocks(old)
