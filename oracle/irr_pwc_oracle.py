"""ORACLE -- test infrastructure, NOT the product.

CPU (torch, fp32) restatement of the IRR-PWC hot path of visinf/irr:
feature pyramid -> warp -> 81-channel cost volume -> shared dense decoders ->
context nets -> bilateral refinement -> occlusion upsampler -> multi-scale loss.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this file.  ``irr_amd`` (the product) never does.

Parity status: PINNED.  ``oracle/gen_golden.py`` imports the reference from
``/root/reference`` in the build container and writes ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks this restatement against those vectors
(eval mode: bit-identical; train mode: losses + per-parameter grad norms).

The restatement is functional (a flat ``{name: tensor}`` parameter dict, the
reference's ``state_dict`` key names) and alias-free: the reference's
``rescale_flow`` mutates its argument (models/pwc_modules.py:70-82) which makes
``IRR_PWC.forward`` double-scale ``flow_cont`` (models/IRR_PWC.py:128-138); here
that observable behaviour is written out explicitly (see ``_level_decode``).

All file:line citations are relative to the reference tree.
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]

SEARCH_RANGE = 4                       # models/IRR_PWC.py:19
NUM_CHS = [3, 16, 32, 64, 96, 128, 196]  # models/IRR_PWC.py:20
OUTPUT_LEVEL = 4                       # models/IRR_PWC.py:21
NUM_LEVELS = 7                         # models/IRR_PWC.py:22
LEVEL_WEIGHTS = [0.32, 0.08, 0.02, 0.01, 0.005, 0.00125, 0.0003125]  # losses.py:522


# --------------------------------------------------------------------------
# parameter inventory: (state_dict prefix, Cin, Cout, k, stride, dilation, lrelu)
# --------------------------------------------------------------------------
def conv_specs() -> List[Tuple[str, int, int, int, int, int, bool]]:
    """Every Conv2d of IRR-PWC in module-definition order (models/IRR_PWC.py:25-46)."""
    s: List[Tuple[str, int, int, int, int, int, bool]] = []
    # FeatureExtractor, models/pwc_modules.py:85-96
    for l, (ci, co) in enumerate(zip(NUM_CHS[:-1], NUM_CHS[1:])):
        s.append((f"feature_pyramid_extractor.convs.{l}.0.0", ci, co, 3, 2, 1, True))
        s.append((f"feature_pyramid_extractor.convs.{l}.1.0", co, co, 3, 1, 1, True))
    dim_corr = (2 * SEARCH_RANGE + 1) ** 2
    ch_flo, ch_occ = dim_corr + 32 + 2, dim_corr + 32 + 1

    def dense(prefix, ch_in, ch_out):  # models/pwc_modules.py:153-161, 190-198
        grow = [0, 128, 256, 352, 416]
        outs = [128, 128, 96, 64, 32]
        for i in range(5):
            s.append((f"{prefix}.conv{i + 1}.0", ch_in + grow[i], outs[i], 3, 1, 1, True))
        s.append((f"{prefix}.conv_last.0", ch_in + 448, ch_out, 3, 1, 1, False))

    def context(prefix, ch_in, ch_out):  # models/pwc_modules.py:210-222, 228-240
        chs = [ch_in, 128, 128, 128, 96, 64, 32]
        dil = [1, 2, 4, 8, 16, 1]
        for i in range(6):
            s.append((f"{prefix}.convs.{i}.0", chs[i], chs[i + 1], 3, 1, dil[i], True))
        s.append((f"{prefix}.convs.6.0", 32, ch_out, 3, 1, 1, False))

    dense("flow_estimators", ch_flo, 2)
    context("context_networks", ch_flo + 448 + 2, 2)
    dense("occ_estimators", ch_occ, 1)
    context("occ_context_networks", ch_occ + 448 + 1, 1)
    # OccUpsampleNetwork(11, 1), models/irr_modules.py:30-44
    s.append(("occ_shuffle_upsample.init_conv.0", 11, 32, 3, 1, 1, True))
    s.append(("occ_shuffle_upsample.res_convs.0.0", 32, 32, 3, 1, 1, True))
    s.append(("occ_shuffle_upsample.res_convs.1.0", 32, 32, 3, 1, 1, False))
    s.append(("occ_shuffle_upsample.res_end_conv.0", 32, 32, 3, 1, 1, True))
    s.append(("occ_shuffle_upsample.out_convs.0", 32, 1, 3, 1, 1, True))
    for i, ci in enumerate([196, 128, 96, 64]):  # models/IRR_PWC.py:38-41
        s.append((f"conv_1x1.{i}.0", ci, 32, 1, 1, 1, True))
    s.append(("conv_1x1_1.0", 16, 3, 1, 1, 1, True))  # models/IRR_PWC.py:43

    def refine(prefix, ch_in):  # models/irr_modules.py:71-79, 115-123
        chs = [ch_in, 128, 128, 64, 64, 32, 32, 9]
        for i in range(7):
            s.append((f"{prefix}.convs.{i}.0", chs[i], chs[i + 1], 3, 1, 1, True))

    refine("refine_flow", 2 + 1 + 32)
    refine("refine_occ", 1 + 32 + 32)
    return s


_SPEC = {p: (ci, co, k, st, dil, act) for (p, ci, co, k, st, dil, act) in conv_specs()}


def synthetic_params(seed: int = 0, dtype=torch.float32) -> Params:
    """Deterministic MSRA-like weights that do not depend on nn.Module RNG order.

    Used by the golden generator (loaded into the reference model) and by tests
    (loaded into the oracle / the HIP model) so no weight file is committed.
    Biases are small non-zero values so the bias path is exercised (the
    reference's own init zeroes them, models/pwc_modules.py:22-27).
    """
    out: Params = {}
    for idx, (prefix, ci, co, k, _st, _dil, _act) in enumerate(conv_specs()):
        g = torch.Generator().manual_seed(seed * 1000003 + idx)
        std = math.sqrt(2.0 / (ci * k * k))
        out[prefix + ".weight"] = (torch.randn(co, ci, k, k, generator=g, dtype=torch.float32) * std).to(dtype)
        out[prefix + ".bias"] = (torch.randn(co, generator=g, dtype=torch.float32) * 0.01).to(dtype)
    return out


# --------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------
def conv_block(p: Params, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """Conv2d(bias, 'same' padding) [+ LeakyReLU(0.1)] -- models/pwc_modules.py:8-19."""
    _ci, _co, k, stride, dil, act = _SPEC[prefix]
    y = F.conv2d(x, p[prefix + ".weight"], p[prefix + ".bias"], stride=stride,
                 padding=((k - 1) * dil) // 2, dilation=dil)
    return F.leaky_relu(y, 0.1) if act else y


def feature_pyramid(p: Params, img: torch.Tensor) -> List[torch.Tensor]:
    """Six (stride-2, stride-1) conv pairs, coarsest level first -- models/pwc_modules.py:98-104."""
    feats = []
    x = img
    for l in range(6):
        x = conv_block(p, f"feature_pyramid_extractor.convs.{l}.0.0", x)
        x = conv_block(p, f"feature_pyramid_extractor.convs.{l}.1.0", x)
        feats.append(x)
    return feats[::-1]


def cost_volume(f1: torch.Tensor, f2: torch.Tensor, max_disp: int = SEARCH_RANGE) -> torch.Tensor:
    """out[b,(dy+r)*D+(dx+r),y,x] = mean_c f1[b,c,y,x]*f2[b,c,y+dy,x+dx], zero outside.

    models/pwc_modules.py:42-62 (Python path actually used) ==
    models/correlation_package/correlation_cuda_kernel.cu:41-114 at
    (pad,k,md,s1,s2)=(4,1,4,1,1).
    """
    _, _, h, w = f1.shape
    r = max_disp
    f2p = F.pad(f2, (r, r, r, r))
    planes = []
    for dy in range(2 * r + 1):          # vertical shift is the slow index (pwc_modules.py:57)
        for dx in range(2 * r + 1):      # horizontal shift is the fast index (pwc_modules.py:58)
            planes.append((f1 * f2p[:, :, dy:dy + h, dx:dx + w]).mean(dim=1, keepdim=True))
    return torch.cat(planes, dim=1)


def correlation_general(f1: torch.Tensor, f2: torch.Tensor, pad_size: int, kernel_size: int, max_displacement: int,
                        stride1: int, stride2: int) -> torch.Tensor:
    """The legacy ``Correlation`` operator at any parameter point, restated from the forward kernel
    (models/correlation_package/correlation_cuda_kernel.cu:41-114; output shape models/correlation_package/correlation_cuda.cc:23-32):
    P = input zero-padded by pad_size, kr = (k - 1) // 2, dr = md // s2, D = 2 dr + 1, window centre (y1, x1) = (oy s1 + md, ox s1 + md),

        out[n, (tj+dr) D + (ti+dr), oy, ox] = 1/(k k C) sum_{j,i in [-kr,kr]} sum_c P1[n,c,y1+j,x1+i] P2[n,c,y1+tj s2+j,x1+ti s2+i]

    (positions outside the padded arrays read as zero).  Differentiable (torch ops only): its autograd gradient is the exact adjoint,
    which for k = 1, s1 = 1 is what correlation_cuda_kernel.cu:116-300 computes.  At (md, 1, md, 1, 1) it equals ``cost_volume`` above
    (= the reference's Python path compute_cost_volume, models/pwc_modules.py:42-62) -- that identity, checked against the imported
    reference, is what pins it; the points with stride2 > 1 / kernel_size > 1 / stride1 > 1 are pinned to a scalar transcription of the
    kernel's loops only (tests/golden/corr_general.npz, oracle/gen_golden.py)."""
    B, C, H, W = f1.shape
    k, md, s1, s2, pad = kernel_size, max_displacement, stride1, stride2, pad_size
    kr, dr = (k - 1) // 2, md // s2
    border = kr + md
    OH = -(-(H + 2 * pad - 2 * border) // s1)
    OW = -(-(W + 2 * pad - 2 * border) // s1)
    # a margin around the padded arrays large enough for every index the loops form (reads as zero)
    m = md + kr + s1
    P1 = F.pad(f1, (pad + m, pad + m, pad + m, pad + m))
    P2 = F.pad(f2, (pad + m, pad + m, pad + m, pad + m))
    ys = torch.arange(OH) * s1 + md + m
    xs = torch.arange(OW) * s1 + md + m
    planes = []
    for tj in range(-dr, dr + 1):
        for ti in range(-dr, dr + 1):
            acc = 0
            for j in range(-kr, kr + 1):
                for i in range(-kr, kr + 1):
                    a = P1[:, :, (ys + j)[:, None], (xs + i)[None, :]]
                    b = P2[:, :, (ys + tj * s2 + j)[:, None], (xs + ti * s2 + i)[None, :]]
                    acc = acc + (a * b).sum(dim=1, keepdim=True)
            planes.append(acc / float(k * k * C))
    return torch.cat(planes, dim=1)


def resize_bilinear_ac(x: torch.Tensor, h: int, w: int) -> torch.Tensor:
    """F.interpolate(bilinear, align_corners=True) -- models/pwc_modules.py:65-67."""
    return F.interpolate(x, [h, w], mode="bilinear", align_corners=True)


def warp(x: torch.Tensor, flow: torch.Tensor, height_im: int, width_im: int, div_flow: float,
         mask_threshold: float = 1.0) -> torch.Tensor:
    """Backward warp + validity mask -- models/pwc_modules.py:107-133.

    Normalisation uses the FULL image size at every level (pwc_modules.py:121-122).
    ``mask_threshold`` = 1.0 is the reference as-is; 0.9999 is the robust-mask
    parity mode of SURVEY.md section 8(c).
    """
    b, _, h, w = x.shape
    gx = torch.linspace(-1.0, 1.0, w).view(1, 1, 1, w).expand(b, 1, h, w)
    gy = torch.linspace(-1.0, 1.0, h).view(1, 1, h, 1).expand(b, 1, h, w)
    base = torch.cat([gx, gy], 1).float()
    fx = flow[:, 0] * 2 / max(width_im - 1, 1) / div_flow
    fy = flow[:, 1] * 2 / max(height_im - 1, 1) / div_flow
    grid = (base + torch.stack([fx, fy], dim=1)).permute(0, 2, 3, 1)
    xw = F.grid_sample(x, grid, align_corners=True)
    ones = torch.ones(x.size())
    m = F.grid_sample(ones, grid, align_corners=True)
    m = (m >= mask_threshold).float()
    return xw * m


def dense_estimator(p: Params, prefix: str, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """DenseNet block, new features PREPENDED -- models/pwc_modules.py:163-170 / 200-207."""
    for i in range(1, 6):
        x = torch.cat([conv_block(p, f"{prefix}.conv{i}.0", x), x], dim=1)
    return x, conv_block(p, f"{prefix}.conv_last.0", x)


def context_net(p: Params, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """7 dilated convs -- models/pwc_modules.py:224-225 / 242-243."""
    for i in range(7):
        x = conv_block(p, f"{prefix}.convs.{i}.0", x)
    return x


def _bilateral_tail(kernel_feat: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """softmax(-f^2) over 9 taps, applied to the replicate-padded 3x3 neighbourhood of every
    channel of ``v`` -- models/irr_modules.py:92-104, 130-139 (tap t = dy*3+dx, Unfold order)."""
    wgt = torch.softmax(-kernel_feat ** 2, dim=1)
    b, c, h, w = v.shape
    vp = F.pad(v, (1, 1, 1, 1), mode="replicate")
    out = torch.zeros_like(v)
    # the reference sums 9 products with torch.sum(dim=1) over the unfolded taps
    taps = []
    for dy in range(3):
        for dx in range(3):
            taps.append(vp[:, :, dy:dy + h, dx:dx + w])
    stacked = torch.stack(taps, dim=2)                      # b, c, 9, h, w
    out = (stacked * wgt.unsqueeze(1)).sum(dim=2)
    return out


def refine_flow(p: Params, flow: torch.Tensor, diff_img: torch.Tensor, feat: torch.Tensor) -> torch.Tensor:
    """models/irr_modules.py:85-104."""
    flow_m = flow - flow.mean(2).mean(2).unsqueeze(2).unsqueeze(2).expand_as(flow)   # :59-60
    nrm = torch.norm(diff_img, p=2, dim=1, keepdim=True)
    x = torch.cat([flow_m, nrm, feat], dim=1)
    for i in range(7):
        x = conv_block(p, f"refine_flow.convs.{i}.0", x)
    return _bilateral_tail(x, flow)


def refine_occ(p: Params, occ: torch.Tensor, feat1: torch.Tensor, feat2: torch.Tensor) -> torch.Tensor:
    """models/irr_modules.py:126-139."""
    x = torch.cat([occ, feat1, feat2], dim=1)
    for i in range(7):
        x = conv_block(p, f"refine_occ.convs.{i}.0", x)
    return _bilateral_tail(x, occ)


def occ_upsample(p: Params, occ: torch.Tensor, guide: torch.Tensor) -> torch.Tensor:
    """models/irr_modules.py:21-27, 46-56.  H, W multiples of 64 => the bilinear fallback never fires."""
    occ = F.interpolate(occ, scale_factor=2, mode="nearest")
    if occ.shape[2:] != guide.shape[2:]:
        occ = F.interpolate(occ, list(guide.shape[2:]), mode="bilinear", align_corners=False)
    pre = "occ_shuffle_upsample"
    x_init = conv_block(p, pre + ".init_conv.0", torch.cat([occ, guide], dim=1))
    x_res = x_init
    for _ in range(3):                                      # shared res_convs, 3x (:51-53)
        r = conv_block(p, pre + ".res_convs.1.0", conv_block(p, pre + ".res_convs.0.0", x_res))
        x_res = x_res + r * 0.1
    x_init = x_init + conv_block(p, pre + ".res_end_conv.0", x_res)
    return conv_block(p, pre + ".out_convs.0", x_init) + occ


# --------------------------------------------------------------------------
# the model -- models/IRR_PWC.py:51-184
# --------------------------------------------------------------------------
def irr_pwc_forward(p: Params, input1: torch.Tensor, input2: torch.Tensor, training: bool,
                    div_flow: float = 0.05, mask_threshold: float = 1.0):
    b, _, H, W = input1.shape
    pyr1 = feature_pyramid(p, input1) + [input1]
    pyr2 = feature_pyramid(p, input2) + [input2]
    h0, w0 = pyr1[0].shape[2:]
    flow_f = torch.zeros(b, 2, h0, w0)
    flow_b = torch.zeros(b, 2, h0, w0)
    occ_f = torch.zeros(b, 1, h0, w0)
    occ_b = torch.zeros(b, 1, h0, w0)
    flows, occs = [], []

    def wl(x, fl):
        return warp(x, fl, H, W, div_flow, mask_threshold)

    for l, (x1, x2) in enumerate(zip(pyr1, pyr2)):
        h, w = x1.shape[2:]
        if l <= OUTPUT_LEVEL:
            if l == 0:
                x2w, x1w = x2, x1
            else:                                            # :82-87
                flow_f = resize_bilinear_ac(flow_f, h, w)
                flow_b = resize_bilinear_ac(flow_b, h, w)
                occ_f = resize_bilinear_ac(occ_f, h, w)
                occ_b = resize_bilinear_ac(occ_b, h, w)
                x2w, x1w = wl(x2, flow_f), wl(x1, flow_b)
            corr_f = F.leaky_relu(cost_volume(x1, x2w), 0.1)  # :90-95
            corr_b = F.leaky_relu(cost_volume(x2, x1w), 0.1)
            if l != OUTPUT_LEVEL:                            # :97-102
                x1p = conv_block(p, f"conv_1x1.{l}.0", x1)
                x2p = conv_block(p, f"conv_1x1.{l}.0", x2)
            else:
                x1p, x2p = x1, x2
            # to_local (:105-106): u *= w/W/div, v *= h/H/div  (pwc_modules.py:72-73)
            s_loc = torch.tensor([float(w / W / div_flow), float(h / H / div_flow)]).view(1, 2, 1, 1)
            # to_global (pwc_modules.py:75-76)
            s_glb = torch.tensor([float(W * div_flow / w), float(H * div_flow / h)]).view(1, 2, 1, 1)
            flow_f = flow_f * s_loc
            flow_b = flow_b * s_loc

            xi_f, res_f = dense_estimator(p, "flow_estimators", torch.cat([corr_f, x1p, flow_f], 1))
            xi_b, res_b = dense_estimator(p, "flow_estimators", torch.cat([corr_b, x2p, flow_b], 1))
            est_f, est_b = flow_f + res_f, flow_b + res_b
            cont_f = est_f + context_net(p, "context_networks", torch.cat([xi_f, est_f], 1))
            cont_b = est_b + context_net(p, "context_networks", torch.cat([xi_b, est_b], 1))

            xo_f, ores_f = dense_estimator(p, "occ_estimators", torch.cat([corr_f, x1p, occ_f], 1))
            xo_b, ores_b = dense_estimator(p, "occ_estimators", torch.cat([corr_b, x2p, occ_b], 1))
            oest_f, oest_b = occ_f + ores_f, occ_b + ores_b
            ocont_f = oest_f + context_net(p, "occ_context_networks", torch.cat([xo_f, oest_f], 1))
            ocont_b = oest_b + context_net(p, "occ_context_networks", torch.cat([xo_b, oest_b], 1))

            img1 = resize_bilinear_ac(input1, h, w)          # :126-127
            img2 = resize_bilinear_ac(input2, h, w)
            # :128-138 with rescale_flow's in-place mutation written out:
            #   G = S*cont is what the image warp AND RefineFlow see; the level's
            #   'flow_cont' output is S*G = S^2*cont; the refined flow is scaled once.
            G_f, G_b = cont_f * s_glb, cont_b * s_glb
            img2w, img1w = wl(img2, G_f), wl(img1, G_b)
            ref_f = refine_flow(p, G_f.detach(), img1 - img2w, x1p)
            ref_b = refine_flow(p, G_b.detach(), img2 - img1w, x2p)
            cont_f, cont_b = G_f * s_glb, G_b * s_glb
            flow_f, flow_b = ref_f * s_glb, ref_b * s_glb

            x2pw, x1pw = wl(x2p, flow_f), wl(x1p, flow_b)     # :141-145
            occ_f = refine_occ(p, ocont_f.detach(), x1p, x1p - x2pw)
            occ_b = refine_occ(p, ocont_b.detach(), x2p, x2p - x1pw)
            flows.append([cont_f, cont_b, flow_f, flow_b])
            occs.append([ocont_f, ocont_b, occ_f, occ_b])
        else:                                                # :150-174
            flow_f = resize_bilinear_ac(flow_f, h, w)
            flow_b = resize_bilinear_ac(flow_b, h, w)
            flows.append([flow_f, flow_b])
            x2w, x1w = wl(x2, flow_f), wl(x1, flow_b)
            fbw, ffw = wl(flow_b, flow_f), wl(flow_f, flow_b)
            if l != NUM_LEVELS - 1:
                a1 = conv_block(p, "conv_1x1_1.0", x1)
                a2 = conv_block(p, "conv_1x1_1.0", x2)
                a1w = conv_block(p, "conv_1x1_1.0", x1w)
                a2w = conv_block(p, "conv_1x1_1.0", x2w)
            else:
                a1, a2, a1w, a2w = x1, x2, x1w, x2w
            occ_f = occ_upsample(p, occ_f, torch.cat([a1, a2w, flow_f, fbw], 1))
            occ_b = occ_upsample(p, occ_b, torch.cat([a2, a1w, flow_b, ffw], 1))
            occs.append([occ_f, occ_b])

    if training:
        return {"flow": flows, "occ": occs}
    return {"flow": resize_bilinear_ac(flow_f, H, W) * (1.0 / div_flow),   # :176-177
            "occ": resize_bilinear_ac(occ_f, H, W)}


# --------------------------------------------------------------------------
# loss -- losses.py:515-577 (MultiScaleEPE_PWC_Bi_Occ_upsample)
# --------------------------------------------------------------------------
def _epe_sum(pred: torch.Tensor, tgt_full: torch.Tensor) -> torch.Tensor:
    tgt = F.adaptive_avg_pool2d(tgt_full, list(pred.shape[2:]))          # losses.py:16-18
    return torch.norm(tgt - pred, p=2, dim=1, keepdim=True).sum()        # losses.py:8-10


def _f1_bal(y_pred: torch.Tensor, y_true: torch.Tensor) -> torch.Tensor:
    """losses.py:39-48."""
    eps = 1e-8
    s = lambda t: t.sum(dim=2).sum(dim=2).sum(dim=1)
    tp = -s(y_true * torch.log(y_pred + eps))
    fn = -s((1 - y_true) * torch.log((1 - y_pred) + eps))
    d_tp = s(y_true) + s(y_pred) + eps
    d_fn = s(1 - y_true) + s(1 - y_pred) + eps
    return ((tp / d_tp).sum() + (fn / d_fn).sum()) * y_pred.size(2) * y_pred.size(3) * 0.5


def multiscale_loss(out: dict, target1, target2, target_occ1, target_occ2, batch_size: int,
                    div_flow: float = 0.05, global_sums=None) -> Dict[str, torch.Tensor]:
    """``global_sums`` = (flow_loss, occ_loss) un-normalised sums over the WHOLE batch when ``out`` holds only a chunk of it
    (every term is a sum over samples, only the balancing weights of losses.py:560-567 couple them): the chunk's
    ``total_loss`` is then its share of the whole batch's, with ``batch_size`` the whole batch's size."""
    tf_f, tf_b = div_flow * target1, div_flow * target2                   # losses.py:535-536
    flow_loss = 0
    occ_loss = 0
    for ii, lvl in enumerate(out["flow"]):                               # losses.py:544-549
        acc = 0
        for jj in range(len(lvl) // 2):
            acc = acc + _epe_sum(lvl[2 * jj], tf_f)
            acc = acc + _epe_sum(lvl[2 * jj + 1], tf_b)
        flow_loss = flow_loss + LEVEL_WEIGHTS[ii] * acc / len(lvl)
    for ii, lvl in enumerate(out["occ"]):                                # losses.py:551-558
        acc = 0
        for jj in range(len(lvl) // 2):
            of, ob = torch.sigmoid(lvl[2 * jj]), torch.sigmoid(lvl[2 * jj + 1])
            acc = acc + _f1_bal(of, F.adaptive_avg_pool2d(target_occ1, list(of.shape[2:])))
            acc = acc + _f1_bal(ob, F.adaptive_avg_pool2d(target_occ2, list(ob.shape[2:])))
        occ_loss = occ_loss + LEVEL_WEIGHTS[ii] * acc / len(lvl)
    fl, ol = (flow_loss.detach(), occ_loss.detach()) if global_sums is None else global_sums   # losses.py:560-567
    if fl > ol:
        w_f, w_o = 1, fl / ol
    else:
        w_f, w_o = ol / fl, 1
    return {"flow_loss": flow_loss / batch_size, "occ_loss": occ_loss / batch_size,
            "total_loss": (flow_loss * w_f + occ_loss * w_o) / batch_size}


def eval_metrics(out: dict, target1, target_occ1) -> Dict[str, torch.Tensor]:
    """losses.py:573-575 with f1_score (losses.py:24-37)."""
    epe = torch.norm(target1 - out["flow"], p=2, dim=1, keepdim=True).mean()
    y_pred = torch.round(torch.sigmoid(out["occ"])).float()
    y_true = target_occ1.float()
    eps = 1e-8
    tp = (y_pred * y_true).sum(dim=2).sum(dim=2)
    prec = tp / (y_pred.sum(dim=2).sum(dim=2) + eps)
    rec = tp / (y_true.sum(dim=2).sum(dim=2) + eps)
    return {"epe": epe, "F1": torch.mean(prec * rec / (prec + rec + eps) * 2)}


# --------------------------------------------------------------------------
# one optimisation step -- runtime.py:131-194 + configuration.py:45-62 + Adam
# (scripts/IRR-PWC_flyingChairsOcc.sh:29-31: lr 1e-4, weight_decay 4e-4)
# --------------------------------------------------------------------------
def synthetic_batch(batch: int, height: int, width: int, seed: int = 1234):
    """SURVEY.md section 8(c) recipe: six draws, in this order, from one generator."""
    g = torch.Generator().manual_seed(seed)
    i1 = torch.rand(batch, 3, height, width, generator=g)
    i2 = torch.rand(batch, 3, height, width, generator=g)
    t1 = 5 * torch.randn(batch, 2, height, width, generator=g)
    t2 = 5 * torch.randn(batch, 2, height, width, generator=g)
    o1 = (torch.rand(batch, 1, height, width, generator=g) < 0.2).float()
    o2 = (torch.rand(batch, 1, height, width, generator=g) < 0.2).float()
    return {"input1": i1, "input2": i2, "target1": t1, "target2": t2,
            "target_occ1": o1, "target_occ2": o2}


def train_step(p: Params, opt: torch.optim.Optimizer, batch: dict, mask_threshold: float = 1.0):
    """zero_grad -> forward -> loss -> backward -> Adam.step; returns python floats."""
    opt.zero_grad()
    out = irr_pwc_forward(p, batch["input1"], batch["input2"], True, mask_threshold=mask_threshold)
    loss = multiscale_loss(out, batch["target1"], batch["target2"], batch["target_occ1"],
                           batch["target_occ2"], batch_size=batch["input1"].shape[0])
    total = loss["total_loss"]
    assert not math.isnan(total.item()), "training_loss is NaN"        # runtime.py:182-183
    total.backward()
    opt.step()
    return {k: float(v.detach()) for k, v in loss.items()}


def train_grads_chunked(p: Params, batch: dict, chunk: int, mask_threshold: float = 1.0) -> Dict[str, float]:
    """Losses and parameter gradients (accumulated into ``p[...].grad``) of ONE batch evaluated ``chunk`` samples at a time --
    identical to the whole-batch step up to summation order, with the memory of a chunk (a 32-pair 384x448 batch holds 53 GB
    of autograd state in one piece).  Pass 1 (no grad): the un-normalised loss sums of the whole batch, which fix the
    balancing weights; pass 2: forward + backward per chunk with those weights."""
    B = batch["input1"].shape[0]
    parts = [{k: v[i:i + chunk] for k, v in batch.items()} for i in range(0, B, chunk)]
    fl = ol = 0.0
    with torch.no_grad():
        for c in parts:
            out = irr_pwc_forward(p, c["input1"], c["input2"], True, mask_threshold=mask_threshold)
            ld = multiscale_loss(out, c["target1"], c["target2"], c["target_occ1"], c["target_occ2"], batch_size=1)
            fl, ol = fl + ld["flow_loss"], ol + ld["occ_loss"]
    total = 0.0
    for c in parts:
        out = irr_pwc_forward(p, c["input1"], c["input2"], True, mask_threshold=mask_threshold)
        ld = multiscale_loss(out, c["target1"], c["target2"], c["target_occ1"], c["target_occ2"], batch_size=B, global_sums=(fl, ol))
        ld["total_loss"].backward()
        total += float(ld["total_loss"].detach())
    return {"flow_loss": float(fl) / B, "occ_loss": float(ol) / B, "total_loss": total}


def make_trainable(p: Params) -> Params:
    return {k: v.clone().requires_grad_(True) for k, v in p.items()}


def make_adam(p: Params) -> torch.optim.Optimizer:
    return torch.optim.Adam(list(p.values()), lr=1e-4, weight_decay=4e-4)
