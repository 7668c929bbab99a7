"""Autograd-visible operators of the IRR-PWC hot path, each a thin shell around one C-ABI call
(include/irr_hip.h).  No CPU path: CPU tensors raise.

Reference call sites: models/IRR_PWC.py:82-95,126-129,141-142,151-158 and the helpers in
models/pwc_modules.py:42-133.
"""
from __future__ import annotations

from typing import Optional

import os
import time

import torch

from . import hip

_GRID_CACHE = {}


def _linspace_dev(n: int, device: torch.device) -> torch.Tensor:
    """torch.linspace(-1, 1, n) computed by torch on the HOST (bit-identical to the reference's
    get_grid, models/pwc_modules.py:107-112) and cached on the device."""
    key = (n, device.index)
    t = _GRID_CACHE.get(key)
    if t is None:
        t = torch.linspace(-1.0, 1.0, n).float().to(device)
        _GRID_CACHE[key] = t
    return t


def _pd(t):
    """the operand as the kernels need it: dense (H, W) planes at a channel stride of H*W, any batch stride (passed as
    ``hip.bs``), 16-byte aligned -- channel or batch slices of a larger tensor qualify and are NOT copied"""
    b, c, h, w = t.shape
    sb, sc, sh, sw = t.stride()
    ok = (sw == 1 or w == 1) and (sh == w or h == 1) and (sc == h * w or c == 1) and (b == 1 or sb % 4 == 0) \
        and t.data_ptr() % 16 == 0
    return t if ok else t.contiguous()


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("irr_amd operators run on the HIP device only (no CPU fallback)")


# ----------------------------------------------------------------------------------------------
# cost volume
# ----------------------------------------------------------------------------------------------
# host times (time.perf_counter) at which the cost-volume GRADIENT launches of the current backward pass were issued -- bench.py relates
# the gradient buckets' all-reduce launches to them (north_star: "all-reduce ... overlapped with the backward correlation kernel");
# cleared by whoever reads it (irr_amd.ddp.GradArena.zero_grad)
CORR_BWD_LAUNCH_TIMES = []


class _CostVolume(hip.Function):
    @staticmethod
    def forward(ctx, f1, f2, lrelu: bool, premasked: bool = False):
        _need_cuda(f1, f2)
        f1, f2 = _pd(f1), _pd(f2)
        B, C, H, W = f1.shape
        out = torch.empty(B, 81, H, W, device=f1.device, dtype=torch.float32)
        hip.call("irr_corr81_fwd_f32", hip.ptr(f1), hip.ptr(f2), hip.ptr(out), B, C, H, W,
                 hip.bs(f1), hip.bs(f2), hip.bs(out), int(lrelu), hip.stream())
        ctx.lrelu = lrelu
        # premasked: every consumer hands back the PRE-activation gradient (conv.dense_estimator(preact_grad_channels=81) applies
        # LeakyReLU' in its data-gradient epilogue), so the 81-plane output is neither kept for nor read by the gradient kernels
        ctx.save_for_backward(f1, f2, out if (lrelu and not premasked) else None)
        return out

    @staticmethod
    def backward(ctx, gout):
        f1, f2, out = ctx.saved_tensors
        gout = _pd(gout)
        B, C, H, W = f1.shape
        g1 = torch.empty_like(f1) if ctx.needs_input_grad[0] else None
        g2 = torch.empty_like(f2) if ctx.needs_input_grad[1] else None
        if len(CORR_BWD_LAUNCH_TIMES) < 64:
            CORR_BWD_LAUNCH_TIMES.append(time.perf_counter())
        hip.call("irr_corr81_bwd_f32", hip.ptr(f1), hip.ptr(f2), hip.ptr(gout), hip.ptr(out), hip.ptr(g1), hip.ptr(g2),
                 B, C, H, W, hip.bs(f1), hip.bs(f2), hip.bs(gout), hip.bs(out) if out is not None else 0,
                 hip.bs(g1) if g1 is not None else 0, hip.bs(g2) if g2 is not None else 0, hip.stream())
        return g1, g2, None, None


def cost_volume(feat1: torch.Tensor, feat2: torch.Tensor, lrelu: bool = False, grad_is_preactivation: bool = False) -> torch.Tensor:
    """81-channel cost volume, optionally with the LeakyReLU(0.1) of models/IRR_PWC.py:94-95 fused.
    grad_is_preactivation (with lrelu): the caller guarantees that EVERY consumer returns the gradient already multiplied by
    LeakyReLU'(out) (conv.dense_estimator(..., preact_grad_channels=81)); the backward kernels then skip the mask and never read
    the 81-plane output (half of their HBM traffic at 96x112)."""
    if feat1.shape != feat2.shape:
        raise ValueError(f"feature maps must have equal shapes, got {tuple(feat1.shape)} vs {tuple(feat2.shape)}")
    if grad_is_preactivation and not lrelu:
        raise ValueError("grad_is_preactivation only makes sense with the fused LeakyReLU")
    return _CostVolume.apply(feat1, feat2, lrelu, bool(grad_is_preactivation))


def compute_cost_volume(feat1, feat2, param_dict):
    """Functional twin with the reference's signature (models/pwc_modules.py:42-62).  The reference reads only ``max_disp`` -- any
    value -- and silently assumes k=1, s1=s2=1: max_disp = 4 runs on the tuned 81-displacement kernels, every other value on the
    general pair (irr_corr_general_*, the same point Correlation(m, 1, m, 1, 1) computes); other kernel sizes / strides are
    rejected here instead of being ignored."""
    for key, want in (("kernel_size", 1), ("stride1", 1), ("stride2", 1)):
        if int(param_dict.get(key, want)) != want:
            raise ValueError(f"{key} must be {want}")
    md = int(param_dict["max_disp"]) if "max_disp" in param_dict else 4
    if md < 0:
        raise ValueError("max_disp must be >= 0")
    if md == 4:
        return cost_volume(feat1, feat2, False)
    from .correlation import _CorrelationGeneral
    return _CorrelationGeneral.apply(feat1, feat2, md, 1, md, 1, 1)


# ----------------------------------------------------------------------------------------------
# warp
# ----------------------------------------------------------------------------------------------
_WARP_BWD_ATOMIC = os.environ.get("IRR_WARP_BWD_ATOMIC", "0") != "0"      # A/B switch: the device-scope atomic scatter everywhere
_WARP_GATHER_MIN_C = int(os.environ.get("IRR_WARP_GATHER_MIN_C", "8"))     # fewer channels: atomic scatter (see _Warp.backward)


class _Warp(hip.Function):
    @staticmethod
    def forward(ctx, x, flow, height_im: int, width_im: int, div_flow: float, mask_thr: float, swap: bool = False):
        _need_cuda(x, flow)
        x, flow = _pd(x), _pd(flow)
        B, C, H, W = x.shape
        gx, gy = _linspace_dev(W, x.device), _linspace_dev(H, x.device)
        out = torch.empty_like(x)
        hip.call("irr_warp_fwd_f32", hip.ptr(x), hip.ptr(flow), hip.ptr(gx), hip.ptr(gy), hip.ptr(out), B, C, H, W,
                 hip.bs(x), hip.bs(flow), hip.bs(out), height_im, width_im, div_flow, mask_thr, int(swap), hip.stream())
        ctx.cfg = (height_im, width_im, div_flow, mask_thr, swap)
        ctx.save_for_backward(x, flow)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, flow = ctx.saved_tensors
        height_im, width_im, div_flow, mask_thr, swap = ctx.cfg
        gout = _pd(gout)
        B, C, H, W = x.shape
        gxg, gyg = _linspace_dev(W, x.device), _linspace_dev(H, x.device)
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gf = torch.empty_like(flow) if ctx.needs_input_grad[1] else None
        # Owner-computes gradient w.r.t. x (csrc/warp.hip: no device-scope atomics; per-sample fallback inside the call) for the
        # feature warps.  Its per-tile binning is paid once per block whatever C is, so the 2- and 3-channel warps (flow, image)
        # keep the one-pass atomic scatter: measured per dispatch in a BASELINE step (noisy flows of a fresh network), 384x448 x 64:
        # C = 3 / 2 gather route 0.68 / 0.58 ms vs atomic 0.49 / 0.33 ms; 192x224 C = 16: 0.39 vs 0.68 ms; 96x112 C = 32: 0.25 vs 0.40.
        if not _WARP_BWD_ATOMIC and (gx is None or C >= _WARP_GATHER_MIN_C):
            ws = torch.empty(hip.lib().irr_warp_bwd_ws_elems(B, H, W), device=x.device, dtype=torch.int32) if gx is not None else None
            hip.call("irr_warp_bwd_gather_f32", hip.ptr(x), hip.ptr(flow), hip.ptr(gxg), hip.ptr(gyg), hip.ptr(gout),
                     hip.ptr(gx), hip.ptr(gf), B, C, H, W, hip.bs(x), hip.bs(flow), hip.bs(gout),
                     hip.bs(gx) if gx is not None else 0, hip.bs(gf) if gf is not None else 0,
                     height_im, width_im, div_flow, mask_thr, int(swap), hip.ptr(ws), ws.numel() if ws is not None else 0, hip.stream())
        else:
            hip.call("irr_warp_bwd_f32", hip.ptr(x), hip.ptr(flow), hip.ptr(gxg), hip.ptr(gyg), hip.ptr(gout),
                     hip.ptr(gx), hip.ptr(gf), B, C, H, W, hip.bs(x), hip.bs(flow), hip.bs(gout),
                     hip.bs(gx) if gx is not None else 0, hip.bs(gf) if gf is not None else 0,
                     height_im, width_im, div_flow, mask_thr, int(swap), hip.stream())
        return gx, gf, None, None, None, None, None


def warp(x, flow, height_im: int, width_im: int, div_flow: float, mask_threshold: float = 1.0, swap_halves: bool = False):
    """WarpingLayer.forward (models/pwc_modules.py:119-133).  swap_halves: sample b warps x[(b + B/2) % B] (the other
    image of a [x1; x2] batch) without a swapped copy of x."""
    if x.shape[0] != flow.shape[0] or x.shape[2:] != flow.shape[2:] or flow.shape[1] != 2:
        raise ValueError(f"bad shapes for warp: x {tuple(x.shape)} flow {tuple(flow.shape)}")
    return _Warp.apply(x, flow, int(height_im), int(width_im), float(div_flow), float(mask_threshold), bool(swap_halves))


# ----------------------------------------------------------------------------------------------
# bilinear resize, align_corners=True
# ----------------------------------------------------------------------------------------------
class _ResizeAC(hip.Function):
    @staticmethod
    def forward(ctx, x, oh: int, ow: int, alpha: float, mode: str = "ac"):
        _need_cuda(x)
        x = _pd(x)
        B, C, H, W = x.shape
        out = torch.empty(B, C, oh, ow, device=x.device, dtype=torch.float32)
        hip.call(f"irr_resize_bilinear_{mode}_fwd_f32", hip.ptr(x), hip.ptr(out), B, C, H, W, oh, ow,
                 hip.bs(x), hip.bs(out), alpha, hip.stream())
        ctx.cfg = (H, W, oh, ow, alpha, mode)
        return out

    @staticmethod
    def backward(ctx, gout):
        H, W, oh, ow, alpha, mode = ctx.cfg
        gout = _pd(gout)
        B, C = gout.shape[:2]
        gx = torch.empty(B, C, H, W, device=gout.device, dtype=torch.float32)
        hip.call(f"irr_resize_bilinear_{mode}_bwd_f32", hip.ptr(gout), hip.ptr(gx), B, C, H, W, oh, ow,
                 hip.bs(gout), hip.bs(gx), alpha, hip.stream())
        return gx, None, None, None, None


class _ResizeACMulti(hip.Function):
    """x resized (align_corners=True) to SEVERAL sizes by one node: forward = the plain launches, backward = ONE gradient buffer that
    the first size's launch overwrites and the others accumulate into (irr_resize_bilinear_ac_bwd_acc_f32) -- as separate nodes the
    autograd engine adds their full-size gradients pairwise.  The raw images at the refinement levels (models/IRR_PWC.py:126-127): five
    downsamplings of a (2B, 3, H, W) tensor whose gradients are 75-99.9 % zeros."""

    @staticmethod
    def forward(ctx, x, *sizes):
        _need_cuda(x)
        x = _pd(x)
        B, C, H, W = x.shape
        outs = []
        for oh, ow in sizes:
            out = torch.empty(B, C, oh, ow, device=x.device, dtype=torch.float32)
            hip.call("irr_resize_bilinear_ac_fwd_f32", hip.ptr(x), hip.ptr(out), B, C, H, W, oh, ow, hip.bs(x), hip.bs(out), 1.0, hip.stream())
            outs.append(out)
        ctx.cfg = (B, C, H, W, tuple(sizes))
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        B, C, H, W, sizes = ctx.cfg
        gx, first = None, True
        for (oh, ow), g in zip(sizes, gouts):
            if g is None:
                continue
            g = _pd(g)
            if gx is None:
                gx = torch.empty(B, C, H, W, device=g.device, dtype=torch.float32)
            hip.call("irr_resize_bilinear_ac_bwd_acc_f32", hip.ptr(g), hip.ptr(gx), B, C, H, W, oh, ow, hip.bs(g), hip.bs(gx), 1.0,
                     0 if first else 1, hip.stream())
            first = False
        return (gx,) + (None,) * len(sizes)


def resize_bilinear_ac_multi(x, sizes):
    """[F.interpolate(x, [oh, ow], mode='bilinear', align_corners=True) for (oh, ow) in sizes] as one autograd node (one gradient buffer)"""
    sizes = tuple((int(h), int(w)) for h, w in sizes)
    if not x.requires_grad or len(sizes) < 2:
        return [resize_bilinear_ac(x, h, w) for h, w in sizes]
    return list(_ResizeACMulti.apply(x, *sizes))


def resize_bilinear_ac(x, oh: int, ow: int, alpha: float = 1.0):
    """alpha * F.interpolate(x, [oh, ow], mode='bilinear', align_corners=True)."""
    return _ResizeAC.apply(x, int(oh), int(ow), float(alpha), "ac")


def resize_bilinear(x, oh: int, ow: int):
    """F.interpolate(x, [oh, ow], mode='bilinear', align_corners=False) (half-pixel centres)."""
    return _ResizeAC.apply(x, int(oh), int(ow), 1.0, "hp")


def upsample_factor2(inputs, target_as):
    """models/irr_modules.py:21-27: nearest x2; when that is not the guide's size (odd pyramid sizes), bilinear
    (align_corners=False) to the guide's size."""
    up = upsample_nearest2x(inputs)
    h, w = target_as.shape[2], target_as.shape[3]
    if up.shape[2] != h or up.shape[3] != w:
        up = resize_bilinear(up, h, w)
    return up


def upsample2d_as(inputs, target_as, mode="bilinear"):
    """models/pwc_modules.py:65-67."""
    if mode != "bilinear":
        raise ValueError("only bilinear is implemented")
    return resize_bilinear_ac(inputs, target_as.shape[2], target_as.shape[3])


# ----------------------------------------------------------------------------------------------
# bilateral refinement tail
# ----------------------------------------------------------------------------------------------
class _RefineTail(hip.Function):
    @staticmethod
    def forward(ctx, feat, v, scale0: float, scale1: float):
        _need_cuda(feat, v)
        feat, v = _pd(feat), _pd(v)
        B, C, H, W = v.shape
        if feat.shape != (B, 9, H, W):
            raise ValueError(f"refine tail expects 9 kernel channels, got {tuple(feat.shape)}")
        out = torch.empty_like(v)
        hip.call("irr_refine_tail_fwd_f32", hip.ptr(feat), hip.ptr(v), hip.ptr(out), B, C, H, W,
                 hip.bs(feat), hip.bs(v), hip.bs(out), scale0, scale1, hip.stream())
        ctx.scales = (scale0, scale1)
        ctx.save_for_backward(feat, v)
        return out

    @staticmethod
    def backward(ctx, gout):
        feat, v = ctx.saved_tensors
        gout = _pd(gout)
        B, C, H, W = v.shape
        gf = torch.empty_like(feat) if ctx.needs_input_grad[0] else None
        gv = torch.empty_like(v) if ctx.needs_input_grad[1] else None
        hip.call("irr_refine_tail_bwd_f32", hip.ptr(feat), hip.ptr(v), hip.ptr(gout), hip.ptr(gf), hip.ptr(gv),
                 B, C, H, W, hip.bs(feat), hip.bs(v), hip.bs(gout), hip.bs(gf) if gf is not None else 0,
                 hip.bs(gv) if gv is not None else 0, ctx.scales[0], ctx.scales[1], hip.stream())
        return gf, gv, None, None


def refine_tail(feat, v, scale=(1.0, 1.0)):
    """scale_c * sum_t softmax_t(-feat^2) * replicate-padded 3x3 neighbourhood of v (models/irr_modules.py:92-104)."""
    return _RefineTail.apply(feat, v, float(scale[0]), float(scale[1]))


# ----------------------------------------------------------------------------------------------
# nearest x2
# ----------------------------------------------------------------------------------------------
class _Nearest2x(hip.Function):
    @staticmethod
    def forward(ctx, x):
        _need_cuda(x)
        x = _pd(x)
        B, C, H, W = x.shape
        out = torch.empty(B, C, 2 * H, 2 * W, device=x.device, dtype=torch.float32)
        hip.call("irr_upsample_nearest2x_fwd_f32", hip.ptr(x), hip.ptr(out), B, C, H, W, hip.bs(x), hip.bs(out), hip.stream())
        return out

    @staticmethod
    def backward(ctx, gout):
        gout = _pd(gout)
        B, C, OH, OW = gout.shape
        gx = torch.empty(B, C, OH // 2, OW // 2, device=gout.device, dtype=torch.float32)
        hip.call("irr_upsample_nearest2x_bwd_f32", hip.ptr(gout), hip.ptr(gx), B, C, OH // 2, OW // 2,
                 hip.bs(gout), hip.bs(gx), hip.stream())
        return gx


def upsample_nearest2x(x):
    """F.interpolate(x, scale_factor=2, mode='nearest') (models/irr_modules.py:22)."""
    return _Nearest2x.apply(x)
