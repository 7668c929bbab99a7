"""GPU parity of the fp32-MFMA conv family against torch's CPU convolution (what the reference executes:
nn.Conv2d + LeakyReLU, models/pwc_modules.py:8-19), forward and all three gradients."""
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
