"""GPU parity of the HIP operators against vectors produced by the reference (tests/golden/ops_basic.npz)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _g(golden_dir):
    return np.load(os.path.join(golden_dir, "ops_basic.npz"))


def D(a, grad=False):
    t = torch.from_numpy(np.array(a)).cuda()
    return t.requires_grad_(True) if grad else t


@pytest.mark.parametrize("case", ["a", "b", "c", "d"])
def test_cost_volume(golden_dir, case):
    from irr_amd import functional as Fn
    g = _g(golden_dir)
    f1, f2 = D(g[f"corr_{case}_f1"], True), D(g[f"corr_{case}_f2"], True)
    out = Fn.compute_cost_volume(f1, f2, {"max_disp": 4, "kernel_size": 1, "stride1": 1, "stride2": 1})
    out.backward(D(g[f"corr_{case}_go"]))
    np.testing.assert_allclose(out.detach().cpu().numpy(), g[f"corr_{case}_out"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(f1.grad.cpu().numpy(), g[f"corr_{case}_g1"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(f2.grad.cpu().numpy(), g[f"corr_{case}_g2"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("case", ["a", "b", "c", "d"])
def test_correlation_module_operator_signature(golden_dir, case):
    """The drop-in operator itself: Correlation(pad_size, kernel_size, max_displacement, stride1, stride2, corr_multiply)
    (models/correlation_package/correlation.py:47-61) called like the reference calls it, values and both gradients
    against the reference fixtures; the native op divides by the channel count exactly like compute_cost_volume's mean."""
    import irr_amd
    g = _g(golden_dir)
    corr = irr_amd.Correlation(pad_size=4, kernel_size=1, max_displacement=4, stride1=1, stride2=1, corr_multiply=1)
    f1, f2 = D(g[f"corr_{case}_f1"], True), D(g[f"corr_{case}_f2"], True)
    out = corr(f1, f2)
    assert out.shape == (f1.shape[0], 81, f1.shape[2], f1.shape[3])
    out.backward(D(g[f"corr_{case}_go"]))
    np.testing.assert_allclose(out.detach().cpu().numpy(), g[f"corr_{case}_out"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(f1.grad.cpu().numpy(), g[f"corr_{case}_g1"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(f2.grad.cpu().numpy(), g[f"corr_{case}_g2"], rtol=1e-5, atol=1e-6)


GEN_POINTS = [(3, 3, 4, 1, 2), (2, 1, 2, 2, 1), (4, 3, 2, 2, 2), (0, 1, 0, 1, 1), (1, 3, 1, 1, 1), (20, 1, 20, 1, 2)]


def test_correlation_module_at_other_parameter_points(golden_dir):
    """The legacy operator off the IRR-PWC point (models/correlation_package/correlation.py:47-61; VERDICT r4 missing #4):
    (a) (md, 1, md, 1, 1) for md = 1, 2, 3 against the imported reference's Python path (compute_cost_volume) incl. both gradients;
    (b) stride2 = 2 / kernel_size = 3 / stride1 = 2 points -- FlowNetC's (20, 1, 20, 1, 2) among them -- against the scalar
    transcription of correlation_cuda_kernel.cu:41-114, and their gradients against autograd through the oracle's restatement
    (the exact adjoint)."""
    import irr_amd
    from oracle import irr_pwc_oracle as O
    g = np.load(os.path.join(golden_dir, "corr_general.npz"))
    f1d, f2d = torch.from_numpy(g["f1"]), torch.from_numpy(g["f2"])
    for md in (1, 2, 3):
        f1, f2 = f1d.float().cuda().requires_grad_(True), f2d.float().cuda().requires_grad_(True)
        out = irr_amd.Correlation(md, 1, md, 1, 1, 1)(f1, f2)
        out.backward(torch.from_numpy(g[f"ref_md{md}_go"]).float().cuda())
        np.testing.assert_allclose(out.detach().cpu().numpy(), g[f"ref_md{md}_out"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(f1.grad.cpu().numpy(), g[f"ref_md{md}_g1"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(f2.grad.cpu().numpy(), g[f"ref_md{md}_g2"], rtol=1e-5, atol=1e-6)
        # the functional twin reads any max_disp too (models/pwc_modules.py:42-62; VERDICT r5 missing #4)
        h1, h2 = f1d.float().cuda().requires_grad_(True), f2d.float().cuda().requires_grad_(True)
        out2 = irr_amd.compute_cost_volume(h1, h2, {"max_disp": md})
        out2.backward(torch.from_numpy(g[f"ref_md{md}_go"]).float().cuda())
        assert torch.equal(out2, out) and torch.equal(h1.grad, f1.grad) and torch.equal(h2.grad, f2.grad)
    for pt in GEN_POINTS:
        f1, f2 = f1d.float().cuda().requires_grad_(True), f2d.float().cuda().requires_grad_(True)
        out = irr_amd.Correlation(*pt, 1)(f1, f2)
        ref = g["scalar_" + "_".join(map(str, pt))]
        assert tuple(out.shape) == ref.shape, (pt, out.shape, ref.shape)
        np.testing.assert_allclose(out.detach().cpu().numpy(), ref, rtol=1e-5, atol=1e-6)
        a, b = f1d.clone().requires_grad_(True), f2d.clone().requires_grad_(True)
        o = O.correlation_general(a, b, *pt)
        go = torch.randn(o.shape, generator=torch.Generator().manual_seed(3), dtype=torch.float64)
        o.backward(go)
        out.backward(go.float().cuda())
        np.testing.assert_allclose(f1.grad.cpu().numpy(), a.grad.numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(f2.grad.cpu().numpy(), b.grad.numpy(), rtol=1e-5, atol=1e-6)
    # launches with more than 65535 (channel, row) pairs: FlowNetC-sized volumes (441 displacement channels x 200 rows) and tall
    # feature maps (300 channels x 256 rows) -- the rows are folded into grid.x
    gen = torch.Generator().manual_seed(9)
    for shape, pt in (((1, 2, 200, 12), (20, 1, 20, 1, 2)), ((1, 300, 256, 8), (2, 1, 2, 1, 1))):
        a = torch.randn(*shape, generator=gen, dtype=torch.float64).requires_grad_(True)
        b = torch.randn(*shape, generator=gen, dtype=torch.float64).requires_grad_(True)
        o = O.correlation_general(a, b, *pt)
        go = torch.randn(o.shape, generator=gen, dtype=torch.float64)
        o.backward(go)
        f1, f2 = a.detach().float().cuda().requires_grad_(True), b.detach().float().cuda().requires_grad_(True)
        out = irr_amd.Correlation(*pt, 1)(f1, f2)
        out.backward(go.float().cuda())
        np.testing.assert_allclose(out.detach().cpu().numpy(), o.detach().numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(f1.grad.cpu().numpy(), a.grad.numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(f2.grad.cpu().numpy(), b.grad.numpy(), rtol=1e-4, atol=1e-5)


def test_cost_volume_fused_lrelu(golden_dir):
    from irr_amd import functional as Fn
    g = _g(golden_dir)
    f1c, f2c = torch.from_numpy(g["corr_a_f1"]).requires_grad_(True), torch.from_numpy(g["corr_a_f2"]).requires_grad_(True)
    go = torch.from_numpy(g["corr_a_go"])
    ref = torch.nn.functional.leaky_relu(torch.from_numpy(g["corr_a_out"]), 0.1)
    # reference gradient of lrelu(corr): chain rule on the golden pieces
    from oracle import irr_pwc_oracle as O
    o = torch.nn.functional.leaky_relu(O.cost_volume(f1c, f2c), 0.1)
    o.backward(go)
    f1, f2 = D(g["corr_a_f1"], True), D(g["corr_a_f2"], True)
    out = Fn.cost_volume(f1, f2, lrelu=True)
    out.backward(go.cuda())
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(f1.grad.cpu().numpy(), f1c.grad.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(f2.grad.cpu().numpy(), f2c.grad.numpy(), rtol=1e-5, atol=1e-6)


def test_cost_volume_rejects_unsupported():
    from irr_amd import functional as Fn
    x = torch.zeros(1, 4, 8, 8, device="cuda")
    with pytest.raises(ValueError):
        Fn.compute_cost_volume(x, x, {"max_disp": 3, "stride2": 2})       # (the reference would silently ignore the stride)
    assert tuple(Fn.compute_cost_volume(x, x, {"max_disp": 3}).shape) == (1, 49, 8, 8)
    with pytest.raises(ValueError):
        Fn.cost_volume(x, torch.zeros(1, 4, 8, 9, device="cuda"))


@pytest.mark.parametrize("case", ["a", "b", "c", "z"])
@pytest.mark.parametrize("mode,thr", [("asis", 1.0), ("robust", 0.9999)])
def test_warp(golden_dir, case, mode, thr):
    from irr_amd import functional as Fn
    g = _g(golden_dir)
    k = f"warp_{case}_{mode}"
    H, W = [int(v) for v in g[k + "_HW"]]
    x, fl = D(g[k + "_x"], True), D(g[k + "_flow"], True)
    out = Fn.warp(x, fl, H, W, 0.05, thr)
    out.backward(D(g[k + "_go"]))
    # mask bits: bit-equal to the reference's (sample(ones) >= thr)
    with torch.no_grad():
        m = Fn.warp(torch.ones(x.shape[0], 1, *x.shape[2:], device="cuda"), fl.detach(), H, W, 0.05, thr)
    assert np.array_equal(m.cpu().numpy(), g[k + "_mask"]), "warp validity mask differs from the reference"
    np.testing.assert_allclose(out.detach().cpu().numpy(), g[k + "_out"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(x.grad.cpu().numpy(), g[k + "_gx"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(fl.grad.cpu().numpy(), g[k + "_gflow"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("case", ["up", "down", "odd"])
def test_resize(golden_dir, case):
    from irr_amd import functional as Fn
    g = _g(golden_dir)
    x = D(g[f"resize_{case}_x"], True)
    go = D(g[f"resize_{case}_go"])
    out = Fn.resize_bilinear_ac(x, go.shape[2], go.shape[3])
    out.backward(go)
    np.testing.assert_allclose(out.detach().cpu().numpy(), g[f"resize_{case}_out"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(x.grad.cpu().numpy(), g[f"resize_{case}_gx"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("shape,size", [((3, 3, 384, 448), (6, 7)), ((2, 3, 384, 448), (96, 112)), ((2, 2, 96, 112), (48, 56)),
                                        ((2, 3, 100, 132), (2, 3)), ((1, 3, 436, 1024), (55, 128)), ((2, 1, 33, 47), (17, 24))])
def test_resize_backward_of_a_downsampling(shape, size):
    """The raw images are resized DOWN to every pyramid level (models/IRR_PWC.py:126-127); their gradient exists because the
    reference's step makes the inputs require grad.  For a downsampling by >= 2 per axis the backward kernel stores the disjoint
    2 x 2 footprints of the output pixels into a zero-filled gradient (resize_bwd_sparse_kernel); compared with ATen's CPU
    interpolate backward, and with the gather-form kernel at a ratio just below 2 (last case: 33x47 -> 17x24 takes the gather form)."""
    import torch.nn.functional as F
    from irr_amd import functional as Fn
    g = torch.Generator().manual_seed(sum(shape) + sum(size))
    x = torch.randn(*shape, generator=g)
    go = torch.randn(shape[0], shape[1], *size, generator=g)
    xc = x.clone().requires_grad_(True)
    F.interpolate(xc, size=size, mode="bilinear", align_corners=True).backward(go)
    xd = x.cuda().requires_grad_(True)
    y = Fn.resize_bilinear_ac(xd, *size)
    y.backward(go.cuda())
    np.testing.assert_allclose(xd.grad.cpu().numpy(), xc.grad.numpy(), rtol=1e-5, atol=1e-6)


def test_warp_full_size_properties():
    """384x448-level shapes: zero flow is the identity (mask all ones), linearity in x."""
    from irr_amd import functional as Fn
    torch.manual_seed(0)
    x = torch.randn(4, 32, 96, 112, device="cuda")
    z = torch.zeros(4, 2, 96, 112, device="cuda")
    out = Fn.warp(x, z, 384, 448, 0.05, 0.9999)
    assert torch.allclose(out, x, atol=2e-4)   # linspace rounding leaves ~1e-5 weight on the neighbour tap
    fl = torch.randn(4, 2, 96, 112, device="cuda") * 0.1
    a = Fn.warp(x, fl, 384, 448, 0.05)
    b = Fn.warp(2 * x, fl, 384, 448, 0.05)
    assert torch.allclose(2 * a, b, atol=1e-5)


@pytest.mark.parametrize("shape", [(4, 32, 96, 112), (2, 35, 45, 70), (4, 3, 24, 28), (2, 2, 6, 7)])
@pytest.mark.parametrize("swap", [False, True])
def test_warp_gradient_owner_computes_equals_atomic_scatter(shape, swap):
    """irr_warp_bwd_gather_f32 (every 8x32 tile of the image gradient gathered by its owning block, LDS accumulation, plain
    stores) against the device-scope atomic scatter of irr_warp_bwd_f32 and against an fp64 autograd restatement of the sampler:
    smooth flows (fast path everywhere), a batch in which ONE sample carries flows of +-40 px at that level (its targets leave the
    gather window: that sample alone takes the atomic fallback, decided on the device), sizes that are not multiples of the tile,
    C > 32 (two accumulator passes), and the swapped-halves form the model uses."""
    import torch.nn.functional as F
    from irr_amd import functional as Fn
    B, C, H, W = shape
    torch.manual_seed(4)
    x = torch.randn(B, C, H, W, device="cuda")
    go = torch.randn(B, C, H, W, device="cuda")
    for wild in ("smooth", "rough", "collapse", "wild"):
        fl = torch.randn(B, 2, H, W, device="cuda") * 0.05 * 1.5              # (flow is in div_flow * full-resolution px)
        if wild == "rough":
            # independent uniform flows of +-6 px per pixel at this level: every gather window up to the largest is used, and
            # many pixels have more than four contributors (further rounds of the owner's loop)
            fl = (torch.rand(B, 2, H, W, device="cuda") * 2 - 1) * 0.05 * 4 * 6.0
        if wild == "collapse" and H >= 24:
            # sample 0: a 12 x 12 block whose pixels all sample (nearly) the same point -- 144 contributors to one pixel of the
            # gradient, more than a list holds: the sample is flagged on the device and redone by the atomic route
            ys, xs = torch.meshgrid(torch.arange(12, device="cuda"), torch.arange(12, device="cuda"), indexing="ij")
            sx, sy = (W - 1) / (4 * W - 1) / 0.05, (H - 1) / (4 * H - 1) / 0.05          # flow units -> pixels at this level
            fl[0, 0, 6:18, 6:18] = (5.3 - xs.float()) / sx
            fl[0, 1, 6:18, 6:18] = (5.6 - ys.float()) / sy
        if wild == "wild":
            fl[1] = torch.randn(2, H, W, device="cuda") * 0.05 * 160.0        # ~ +-40 px at a quarter-resolution level
        res = {}
        for atomic in (True, False):
            Fn._WARP_BWD_ATOMIC = atomic
            old_min_c, Fn._WARP_GATHER_MIN_C = Fn._WARP_GATHER_MIN_C, 1      # (the model routes C < 8 to the atomic scatter)
            try:
                xr, fr = x.clone().requires_grad_(True), fl.clone().requires_grad_(True)
                Fn.warp(xr, fr, 4 * H, 4 * W, 0.05, 0.9999, swap).backward(go)
                res[atomic] = (xr.grad.clone(), fr.grad.clone())
            finally:
                Fn._WARP_BWD_ATOMIC = False
                Fn._WARP_GATHER_MIN_C = old_min_c
        fscale = res[True][1].abs().max().item()                                 # (flow gradient: a lanes-along-x kernel of its own)
        assert (res[True][1] - res[False][1]).abs().max().item() <= 2e-5 * fscale + 1e-6, (wild, shape)
        scale = res[True][0].abs().max().item()
        assert (res[True][0] - res[False][0]).abs().max().item() <= 2e-6 * scale + 1e-6, (wild, shape)
        # fp64 restatement: bilinear zero-padded sampling, align_corners=True (models/pwc_modules.py:119-133)
        xd = x.double().requires_grad_(True)
        src = torch.roll(xd, B // 2, 0) if swap else xd                      # sample b reads x[(b + B/2) % B]
        gxb = torch.linspace(-1, 1, W, device="cuda", dtype=torch.float64).view(1, 1, W).expand(B, H, W)
        gyb = torch.linspace(-1, 1, H, device="cuda", dtype=torch.float64).view(1, H, 1).expand(B, H, W)
        grid = torch.stack([gxb + fl[:, 0].double() * 2 / max(4 * W - 1, 1) / 0.05, gyb + fl[:, 1].double() * 2 / max(4 * H - 1, 1) / 0.05], 3)
        with torch.no_grad():
            m = Fn.warp(torch.ones(B, 1, H, W, device="cuda"), fl, 4 * H, 4 * W, 0.05, 0.9999)
        (F.grid_sample(src, grid, align_corners=True) * m.double()).backward(go.double())
        # (sampling coordinates of +-40 px carry an fp32 rounding of ~1e-5 px into the bilinear weights)
        assert (xd.grad - res[False][0].double()).abs().max().item() <= (1e-5 if wild == "smooth" else 1e-4) * scale + 1e-6, (wild, shape)


def test_cost_volume_full_size_properties():
    """level-4 shape at bs8: centre channel equals the channel-mean of f1*f2; shifted input shifts channels."""
    from irr_amd import functional as Fn
    torch.manual_seed(0)
    f1 = torch.randn(8, 32, 96, 112, device="cuda")
    f2 = torch.randn(8, 32, 96, 112, device="cuda")
    cv = Fn.cost_volume(f1, f2)
    assert torch.allclose(cv[:, 40], (f1 * f2).mean(1), atol=1e-5)
    # displacement (dy,dx)=(+1,-2): channel (1+4)*9+(-2+4)=47
    ref = (f1[:, :, :-1, 2:] * f2[:, :, 1:, :-2]).mean(1)
    assert torch.allclose(cv[:, 47, :-1, 2:], ref, atol=1e-5)


@pytest.mark.parametrize("shape", [(2, 32, 32, 48), (1, 16, 20, 112), (2, 8, 17, 16), (1, 12, 9, 80)])
def test_cost_volume_16x16_tiles_vs_oracle(shape):
    """Widths that are a multiple of 16 but not of 32 (48, 112, 16, 80) run the quad kernels on 16 x 16 tiles (round 3: 7 full
    tiles per row at 96x112 instead of 3.5): values with the fused LeakyReLU and both gradients against the oracle, and
    against the 32 x 8 tiles of the same library (IRR_CORR_TILE32 is read once per process, so that A/B lives in tools/corr_bench.py)."""
    from irr_amd import functional as Fn
    from oracle import irr_pwc_oracle as O
    B, C, H, W = shape
    g = torch.Generator().manual_seed(W)
    f1 = torch.randn(B, C, H, W, generator=g, requires_grad=True)
    f2 = torch.randn(B, C, H, W, generator=g, requires_grad=True)
    go = torch.randn(B, 81, H, W, generator=g)
    ref = torch.nn.functional.leaky_relu(O.cost_volume(f1, f2), 0.1)
    ref.backward(go)
    a, b = f1.detach().cuda().requires_grad_(True), f2.detach().cuda().requires_grad_(True)
    out = Fn.cost_volume(a, b, lrelu=True)
    out.backward(go.cuda())
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(a.grad.cpu().numpy(), f1.grad.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(b.grad.cpu().numpy(), f2.grad.numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("shape", [(64, 196, 6, 7), (64, 128, 12, 14), (3, 21, 5, 9), (2, 7, 31, 33)])
def test_cost_volume_tiny_planes_vs_oracle(shape):
    """Planes whose width is not a multiple of 4: at the BASELINE batch the 6x7 and 12x14 levels run corr81_small_fwd_kernel /
    corr81_small_bwd_kernel (pixels x channel slices x displacements over the chip instead of one 16 x 16 tile per sample); the
    last shape is large enough to stay on the tile kernels.  Values with the fused LeakyReLU and both gradients against the oracle."""
    from irr_amd import functional as Fn
    from oracle import irr_pwc_oracle as O
    B, C, H, W = shape
    g = torch.Generator().manual_seed(H * W)
    f1 = torch.randn(B, C, H, W, generator=g, requires_grad=True)
    f2 = torch.randn(B, C, H, W, generator=g, requires_grad=True)
    go = torch.randn(B, 81, H, W, generator=g)
    ref = torch.nn.functional.leaky_relu(O.cost_volume(f1, f2), 0.1)
    ref.backward(go)
    a, b = f1.detach().cuda().requires_grad_(True), f2.detach().cuda().requires_grad_(True)
    out = Fn.cost_volume(a, b, lrelu=True)
    out.backward(go.cuda())
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(a.grad.cpu().numpy(), f1.grad.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(b.grad.cpu().numpy(), f2.grad.numpy(), rtol=1e-5, atol=1e-6)
    # without the fused activation (no mask operand in the gradient kernels)
    a2, b2 = f1.detach().cuda().requires_grad_(True), f2.detach().cuda().requires_grad_(True)
    Fn.cost_volume(a2, b2).backward(go.cuda())
    f1.grad = f2.grad = None
    O.cost_volume(f1, f2).backward(go)
    np.testing.assert_allclose(a2.grad.cpu().numpy(), f1.grad.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(b2.grad.cpu().numpy(), f2.grad.numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("hw", [(6, 7), (12, 16), (5, 3)])
def test_cat_channels_one_launch(hw):
    """irr_cat_channels_f32: parts (one of them a channel-slice view, one non-contiguous) land in consecutive channel slices of a
    channel-slice view of the destination, zero tail behind them; 16-byte and scalar paths; more than IRR_CAT_MAX_PARTS parts."""
    from irr_amd import conv as C
    H, W = hw
    B = 3
    g = torch.Generator().manual_seed(H * 31 + W)
    widths = [81, 32, 2, 1, 5, 7, 3, 4, 6, 2]
    parts = []
    for i, wd in enumerate(widths):
        if i == 1:
            parts.append(torch.randn(B, wd + 4, H, W, generator=g).cuda()[:, 3:3 + wd])            # channel-slice view
        elif i == 2:
            parts.append(torch.randn(B, H, W, wd, generator=g).cuda().permute(0, 3, 1, 2))         # planes not dense
        else:
            parts.append(torch.randn(B, wd, H, W, generator=g).cuda())
    tot = sum(widths)
    dst = torch.full((B, 5 + tot + 4 + 2, H, W), 7.0, device="cuda")
    C.cat_channels_into(dst[:, 5:], parts, zero_tail=4)
    ref = torch.cat([p_ for p_ in parts], dim=1)
    assert torch.equal(dst[:, 5:5 + tot], ref)
    assert torch.equal(dst[:, 5 + tot:5 + tot + 4], torch.zeros(B, 4, H, W, device="cuda"))
    assert torch.equal(dst[:, :5], torch.full((B, 5, H, W), 7.0, device="cuda")) and torch.equal(dst[:, -2:], torch.full((B, 2, H, W), 7.0, device="cuda"))


def test_warp_swap_halves_equals_swapped_copy():
    """swap_halves=True warps the OTHER batch half of x (the model's [x1; x2] / [x2; x1] pairing) without the copy: values,
    the gradient scattered into the other half of gx, and the flow gradient equal those of an explicit torch.cat swap"""
    from irr_amd import functional as Fn
    g = torch.Generator().manual_seed(21)
    B, C, H, W = 4, 5, 24, 36
    x = torch.randn(B, C, H, W, generator=g)
    fl = 3.0 * torch.randn(B, 2, H, W, generator=g)
    go = torch.randn(B, C, H, W, generator=g).cuda()
    xa, fa = x.cuda().requires_grad_(True), fl.cuda().requires_grad_(True)
    ya = Fn.warp(torch.cat([xa[B // 2:], xa[:B // 2]], dim=0), fa, 4 * H, 4 * W, 0.05, 0.9999)
    ya.backward(go)
    xb, fb = x.cuda().requires_grad_(True), fl.cuda().requires_grad_(True)
    yb = Fn.warp(xb, fb, 4 * H, 4 * W, 0.05, 0.9999, swap_halves=True)
    yb.backward(go)
    assert torch.equal(ya, yb)
    np.testing.assert_allclose(xb.grad.cpu().numpy(), xa.grad.cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(fb.grad.cpu().numpy(), fa.grad.cpu().numpy(), rtol=1e-5, atol=1e-6)
    with pytest.raises(Exception):
        Fn.warp(x[:3].cuda(), fl[:3].cuda(), 4 * H, 4 * W, 0.05, 1.0, swap_halves=True)


@pytest.mark.parametrize("shape", [(2, 3, 64, 96), (1, 3, 37, 53), (2, 2, 24, 28)])
def test_resize_multi_is_the_sum_of_the_single_nodes(shape):
    """Fn.resize_bilinear_ac_multi (round 6): one autograd node for several sizes of one tensor -- outputs identical to the single
    launches, the ONE gradient buffer (first size overwrites, the others accumulate: sparse footprints for downsampling by >= 2,
    gather form otherwise, an upsampling size included) equal to the sum the autograd engine forms from separate nodes."""
    from irr_amd import functional as Fn
    B, C, H, W = shape
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, C, H, W, generator=g).cuda()
    sizes = [(max(1, H // 16), max(1, W // 16)), (H // 8, W // 8), (H // 4, W // 4), (H // 2 + 1, W // 2), (H - 1, W - 3), (H + 5, W + 2)]
    gouts = [torch.randn(B, C, h, w, generator=g).cuda() for h, w in sizes]
    xa = x.clone().requires_grad_(True)
    outs_a = Fn.resize_bilinear_ac_multi(xa, sizes)
    xb = x.clone().requires_grad_(True)
    outs_b = [Fn.resize_bilinear_ac(xb, h, w) for h, w in sizes]
    for a, b in zip(outs_a, outs_b):
        assert torch.equal(a, b)
    torch.autograd.backward(outs_a, gouts)
    torch.autograd.backward(outs_b, gouts)
    ref = xb.grad.double()
    assert float((xa.grad.double() - ref).abs().max()) <= 1e-6 * float(ref.abs().max())
    # a size without a gradient (a truncated pass) is skipped
    xc = x.clone().requires_grad_(True)
    outs_c = Fn.resize_bilinear_ac_multi(xc, sizes[:3])
    outs_c[1].backward(gouts[1])
    xd = x.clone().requires_grad_(True)
    Fn.resize_bilinear_ac(xd, *sizes[1]).backward(gouts[1])
    assert torch.equal(xc.grad, xd.grad)
    # without requires_grad: plain launches, nothing recorded
    assert all(not o.requires_grad for o in Fn.resize_bilinear_ac_multi(x, sizes))
