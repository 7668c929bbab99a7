"""Autograd nodes over the conv primitives of irr_amd.conv: one conv() block, the whole DenseNet estimator, a sequential chain of
conv() blocks and the OccUpsampleNetwork, each as ONE node whose backward uses the epilogue features of the data-gradient launches
(skip adds, 0.1 x, LeakyReLU', accumulate) instead of elementwise passes, and routes weight gradients to the lane when one is
installed (``irr_amd.conv.SIDE``)."""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

from . import conv as _c
from . import hip
from .conv import Amax, _call_conv, _h2_args, amax_measure, conv_dgrad, conv_forward, conv_forward_skip, conv_wgrad, h2_code, x3_code
from .conv_pack import LAUNCHES, _dense_column_packs, _padded_cin


def wgrad_param(x, gy, weight, bias, stride: int, dil: int, want_bias: bool = True, alpha: float = 1.0,
                acc=None, x_amax: Optional[Amax] = None, gy_amax: Optional[Amax] = None, x_chmax=None, gy_chmax=None):
    """Weight (+bias) gradient of one conv use.  Returns (gw, gb) tensors for autograd -- or (None, None) when the
    result was accumulated asynchronously into the gradient arena (SIDE lane).  ``acc`` = optional (gw, gb) pair to
    accumulate into (shared weights used several times inside one autograd node).
    x_amax / gy_amax: the operands' amax slots (both given + MATH == "h2": the launch runs on the fp16x2 form)."""
    routed = _c.SIDE.route(weight, bias) if _c.SIDE is not None else None
    if routed is not None:
        gwv, gbv = routed
        keep = (x, gy) if (x_amax is None or gy_amax is None) else (x, gy, x_amax.slots, gy_amax.slots)
        keep = keep + tuple(t_ for t_ in (x_chmax, gy_chmax) if torch.is_tensor(t_))

        def fn():
            if _c._CHECK_FINITE == "slots" and x_amax is not None and gy_amax is not None:
                # diagnosis (NOTES C.5 / D.2): what THIS launch sees on the stream it runs on -- operand magnitudes against their slots
                inf = float("inf")
                for tag, t, a in (("x", x, x_amax), ("gy", gy, gy_amax)):
                    _c._FINITE_LOG.append((f"wgrad {tag} {tuple(t.shape)} -> w {tuple(weight.shape)}", torch.linalg.vector_norm(t, ord=inf),
                                           a.slots[a.first:a.first + a.n].max()))
            conv_wgrad(x, gy, weight.shape, stride, dil, gw=gwv, gbias=gbv if want_bias else None,
                       alpha=alpha, defer=_c.SIDE.batch, x_amax=x_amax, gy_amax=gy_amax, x_chmax=x_chmax, gy_chmax=gy_chmax)

        _c.SIDE.launch(fn, keep, (weight, bias if (want_bias and gbv is not None) else None), gw=gwv)
        return None, None
    if acc is not None:
        gw, gb = acc
    else:
        gw = None
        gb = torch.zeros(weight.shape[0], device=x.device, dtype=torch.float32) if (want_bias and bias is not None) else None
    gw = conv_wgrad(x, gy, weight.shape, stride, dil, gw=gw, gbias=gb if want_bias else None, alpha=alpha,
                    x_amax=x_amax, gy_amax=gy_amax, x_chmax=x_chmax, gy_chmax=gy_chmax)
    return gw, gb


def _fwd_h2(x, weight, stride: int, dil: int) -> bool:
    """conv(x, weight) runs on the fp16x2 form"""
    B, cin, H, W = x.shape
    return bool(h2_code(B, cin, H, W, weight.shape[0], weight.shape[2], stride, dil))


def _dgrad_h2(gy_shape, weight, stride: int, dil: int) -> bool:
    """the stride-1 data gradient of conv(., weight) for an output gradient of gy_shape runs on the fp16x2 form"""
    B, cout, oh, ow = gy_shape
    return stride == 1 and cout >= 2 and bool(h2_code(B, cout, oh, ow, weight.shape[1], weight.shape[2], 1, dil))


def lrelu_bwd_bias(gy: torch.Tensor, y: Optional[torch.Tensor], lrelu: bool, gpre: Optional[torch.Tensor],
                   gbias: Optional[torch.Tensor], amax: Optional[Amax] = None) -> None:
    """gpre = gy * LeakyReLU'(y) (y = the activated output); gbias += sum over (b, h, w) of gpre; amax: a slot that receives
    max |gpre| from the same pass."""
    B, C, H, W = gy.shape
    hip.call("irr_lrelu_bwd_bias_f32", hip.ptr(gy), hip.ptr(y) if lrelu else None, hip.ptr(gpre), hip.ptr(gbias),
             B, C, H * W, hip.bs(gy), hip.bs(y) if lrelu else 0, hip.bs(gpre) if gpre is not None else 0,
             int(lrelu), amax.ptr() if amax is not None else None, hip.stream())


# ----------------------------------------------------------------------------------------------
# autograd: one conv() block
# ----------------------------------------------------------------------------------------------
class _ConvBlock(hip.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride: int, dil: int, lrelu: bool, res, alpha: float):
        if not x.is_cuda:
            raise RuntimeError("irr_amd conv runs on the HIP device only (no CPU fallback)")
        x = x if _planes_dense(x) else x.contiguous()
        if res is not None and not _planes_dense(res):
            res = res.contiguous()
        xa = amax_measure(x) if _fwd_h2(x, weight, stride, dil) else None
        if res is None and alpha == 1.0:
            y = conv_forward(x, weight, bias, stride, dil, lrelu, x_amax=xa)
            act = y
        else:
            # keep the activated conv output for the LeakyReLU derivative
            act = conv_forward(x, weight, bias, stride, dil, lrelu, x_amax=xa)
            y = act * alpha if res is None else torch.add(res, act, alpha=alpha)
        ctx.x_amax = xa
        ctx.cfg = (stride, dil, lrelu, alpha, res is not None)
        ctx.save_for_backward(x, weight, act if lrelu else None)
        ctx.has_bias = bias is not None
        ctx.weight_obj = weight            # the Parameter object that carries the packed-weight cache
        ctx.bias_obj = bias
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, act = ctx.saved_tensors
        stride, dil, lrelu, alpha, has_res = ctx.cfg
        gy = gy if _planes_dense(gy) else gy.contiguous()
        # gy is also read by the asynchronous wgrad lane: hand autograd its own copy, because the engine may
        # accumulate further gradients of `res` into the returned tensor IN PLACE on the main stream
        gres = gy.clone() if (has_res and ctx.needs_input_grad[6]) else None
        g = gy if alpha == 1.0 else gy * alpha
        cout = weight.shape[0]
        gb = torch.zeros(cout, device=gy.device, dtype=torch.float32) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        want_w = ctx.needs_input_grad[1]
        bias_in_wgrad = gb is not None and want_w and not lrelu      # no elementwise pass needed at all
        need_ga = (ctx.x_amax is not None and want_w) or (ctx.needs_input_grad[0] and _dgrad_h2(g.shape, weight, stride, dil))
        ga = None
        if lrelu or (gb is not None and not bias_in_wgrad):
            gpre = torch.empty_like(g) if lrelu else None
            if need_ga:
                ga = Amax.zeros(g.device, 1)                          # max |pre-activation gradient|: folded by the same pass
            lrelu_bwd_bias(g, act, lrelu, gpre, gb, amax=ga)          # mask and bias gradient in one HBM pass
            if lrelu:
                g = gpre
        if need_ga and ga is None:
            ga = amax_measure(g)
        gx = conv_dgrad(g, ctx.weight_obj, stride, dil, x.shape[2:], gy_amax=ga) if ctx.needs_input_grad[0] else None
        gw = None
        if want_w:
            if bias_in_wgrad:
                gw, gb = wgrad_param(x, g, ctx.weight_obj, ctx.bias_obj, stride, dil, want_bias=True, x_amax=ctx.x_amax, gy_amax=ga)
            else:
                gw, _ = wgrad_param(x, g, ctx.weight_obj, None, stride, dil, want_bias=False, x_amax=ctx.x_amax, gy_amax=ga)
        return gx, gw, gb, None, None, None, gres, None


class _CatPart(ctypes.Structure):
    """IrrCatPart of include/irr_hip.h"""
    _fields_ = [("src", ctypes.c_void_p), ("src_bs", ctypes.c_long), ("channels", ctypes.c_int), ("reserved", ctypes.c_int)]


CAT_MAX_PARTS = 8            # IRR_CAT_MAX_PARTS


def cat_channels_into(dst: torch.Tensor, parts, zero_tail: int = 0, amax: Optional[Amax] = None, chmax: Optional[torch.Tensor] = None) -> None:
    """dst[:, :sum(channels)] = cat(parts, dim=1) (+ ``zero_tail`` zero channels behind them) in ONE launch
    (irr_cat_channels_f32) -- dst is a channel-slice view of the consumer's buffer.  Parts whose planes are not dense are made
    contiguous first.  amax: a slot that receives max |.| of everything written (irr_cat_channels_amax_f32: the copy folds the
    magnitude the consumer's fp16x2 kernels need -- no separate pass over the buffer)."""
    B, _, H, W = dst.shape
    srcs = [p_ if _planes_dense(p_) else p_.contiguous() for p_ in parts]
    recs = [(hip.ptr(p_), p_.stride(0), int(p_.shape[1])) for p_ in srcs]
    if zero_tail > 0:
        recs.append((None, 0, int(zero_tail)))
    c0 = 0
    for i in range(0, len(recs), CAT_MAX_PARTS):
        chunk = recs[i:i + CAT_MAX_PARTS]
        arr = (_CatPart * len(chunk))(*[_CatPart(s_, bs_, ch_, 0) for s_, bs_, ch_ in chunk])
        view = dst[:, c0:]
        if chmax is not None and os.environ.get("IRR_CAT_NO_CH_FOLD"):      # (A/B: a pass over what the copy wrote)
            cat_channels_into(dst, parts, zero_tail, amax)
            _c.channel_amax(dst[:, :chmax.numel()], chmax)
            return
        if chmax is not None:                                  # (zeroed slots, one per channel written: the copy folds max |.| per channel too)
            hip.call("irr_cat_channels_amax_ch_f32", hip.ptr(view), dst.stride(0), ctypes.addressof(arr), len(chunk), B, H * W,
                     amax.ptr() if amax is not None else None, hip.ptr(chmax[c0:]), hip.stream())
        elif amax is not None:
            hip.call("irr_cat_channels_amax_f32", hip.ptr(view), dst.stride(0), ctypes.addressof(arr), len(chunk), B, H * W, amax.ptr(),
                     hip.stream())
        else:
            hip.call("irr_cat_channels_f32", hip.ptr(view), dst.stride(0), ctypes.addressof(arr), len(chunk), B, H * W, hip.stream())
        c0 += sum(ch_ for _, _, ch_ in chunk)


class _CatChannelsFn(hip.Function):
    """torch.cat(parts, dim=1) as ONE launch that also folds max |.| of what it writes into an amax slot (the input magnitude of the
    fp16x2 conv chain that consumes the result: RefineFlow / RefineOcc, models/irr_modules.py:97-99, 134); backward hands out the
    channel slices of the incoming gradient (views, no copies)."""

    @staticmethod
    def forward(ctx, slot, *parts):
        B, _, H, W = parts[0].shape
        widths = [int(p_.shape[1]) for p_ in parts]
        out = torch.empty(B, sum(widths), H, W, device=parts[0].device, dtype=torch.float32)
        cat_channels_into(out, parts, amax=slot)
        ctx.widths = widths
        return out

    @staticmethod
    def backward(ctx, g):
        outs, c0 = [None], 0
        for i, wd in enumerate(ctx.widths):
            outs.append(g[:, c0:c0 + wd] if ctx.needs_input_grad[1 + i] else None)
            c0 += wd
        return tuple(outs)


def cat_channels(parts, want_amax: bool = False) -> torch.Tensor:
    """cat(parts, dim=1) on the device in one launch; want_amax: the result carries its magnitude (``_irr_amax``, read by conv_chain)"""
    parts = tuple(parts)
    if not all(p_.is_cuda and p_.dtype == torch.float32 for p_ in parts):
        raise RuntimeError("irr_amd conv runs on the HIP device only (no CPU fallback)")
    slot = Amax.zeros(parts[0].device, 1) if want_amax else None
    out = _CatChannelsFn.apply(slot, *parts)
    if slot is not None:
        out.__dict__["_irr_amax"] = slot
    return out


def add_planes(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """x + y for two (B, C, H, W) tensors with dense planes, either of which may be a channel-slice view (irr_add_planes_f32: one
    coalesced pass; ATen runs such a view through its strided-iterator kernel at a fraction of the bandwidth).  No autograd: for
    the backward passes of the nodes below."""
    if not (_planes_dense(x) and _planes_dense(y) and x.shape == y.shape):
        return x + y
    B, C, H, W = x.shape
    out = torch.empty(B, C, H, W, device=x.device, dtype=torch.float32)
    hip.call("irr_add_planes_f32", hip.ptr(out), hip.ptr(x), hip.ptr(y), B, C * H * W, hip.bs(out), hip.bs(x), hip.bs(y), hip.stream())
    return out


def _planes_dense(t: torch.Tensor) -> bool:
    b, c, h, w = t.shape
    sb, sc, sh, sw = t.stride()
    return (sw == 1 or w == 1) and (sh == w or h == 1) and (sc == h * w or c == 1)


def conv_block(x, weight, bias, stride: int = 1, dil: int = 1, lrelu: bool = True, res=None, alpha: float = 1.0):
    """[res +] alpha * LeakyReLU?(conv2d(x, weight, bias, stride, 'same' padding, dil))."""
    return _ConvBlock.apply(x, weight, bias, int(stride), int(dil), bool(lrelu), res, float(alpha))


# ----------------------------------------------------------------------------------------------
# autograd: the whole DenseNet estimator (FlowEstimatorDense / OccEstimatorDense) as ONE node
# ----------------------------------------------------------------------------------------------
class _DenseEstimatorFn(hip.Function):
    """conv1..conv5 (+LeakyReLU, outputs PREPENDED) and conv_last of models/pwc_modules.py:153-170 / 190-207
    on ONE preallocated NCHW buffer: every conv reads a channel suffix and writes the slice in front of it,
    so there is no torch.cat; the backward walks the same buffer layout with a gradient buffer G in which
    data-gradients are accumulated in place (``accumulate`` epilogue of the MFMA kernel).

    Buffer layout (channels): [c5 32 | c4 64 | c3 96 | c2 128 | c1 128 | x Cin0 | est E]   (est only if base given)
    Returns (buf, out): out = conv_last(x5) (+ base when given; then also stored in the est slot so the
    context network can consume ``buf`` directly as cat([x5, est]), models/IRR_PWC.py:113-114)."""

    GROW = (128, 128, 96, 64, 32)

    @staticmethod
    def forward(ctx, nparts, base, nrelu, *args):
        # x arrives as `nparts` tensors (IRR-PWC: cost volume, projected features, flow / occlusion): they are copied straight
        # into their channel slices of the buffer, and backward returns the slices of the gradient buffer -- no torch.cat of the
        # decoder input in forward, no split of its gradient in backward
        parts, wb = args[:nparts], args[nparts:]
        ws, bs = wb[0::2], wb[1::2]
        B, _, H, W = parts[0].shape
        widths = [int(p_.shape[1]) for p_ in parts]
        cin0 = sum(widths)
        E = ws[5].shape[0]
        ctot = 448 + cin0
        has_base = base is not None
        buf = torch.empty(B, ctot + (E if has_base else 0), H, W, device=parts[0].device, dtype=torch.float32)
        # fp16x2 route: one amax slot per buffer part, in channel order [c5, c4, c3, c2, c1, x]; conv i+1 reads parts 5-i .. 5 and
        # its launch folds the magnitude of its output into slot 4-i; the input part's comes out of the copy that assembles it
        S = None
        if _fwd_h2(buf[:, 448:ctot], ws[0], 1, 1):
            S = Amax.zeros(buf.device, 7)                        # (slot 6: the est slot behind the parts, when there is one)
        # round 6: max |buf[:, c]| per channel, for the weight gradients whose launch takes the buffer as the operand in its gy role
        # (conv3 / conv5: exchanged roles) -- every layer's launch folds its own slice, the copy that assembles the input part folds that one
        Bch = None
        if S is not None and _c.WGRAD_CHANNEL_SCALE and any(ctx.needs_input_grad[3 + nparts:]):      # (a training pass: some weight wants its gradient)
            Bch = _c.zero_slots(buf.device, ctot)
        cat_channels_into(buf[:, 448:], parts, amax=S.sub(5) if S is not None else None, chmax=Bch[448:] if Bch is not None else None)
        off = 448
        for i in range(5):
            co = _DenseEstimatorFn.GROW[i]
            conv_forward(buf[:, off:ctot], ws[i], bs[i], 1, 1, True, out=buf[:, off - co:off],
                         x_amax=S.sub(5 - i, i + 1) if S is not None else None, y_amax=S.sub(4 - i) if S is not None else None,
                         y_chmax=Bch[off - co:off] if Bch is not None else None)
            off -= co
        ctx.amax = S
        ctx.bch = Bch
        if has_base:
            base_c = base if _planes_dense(base) else base.contiguous()
            out = conv_forward(buf[:, :ctot], ws[5], bs[5], 1, 1, False, res=base_c, alpha=1.0)
            buf[:, ctot:].copy_(out)
            if S is not None:
                amax_measure(out, S.sub(6))
        else:
            out = conv_forward(buf[:, :ctot], ws[5], bs[5], 1, 1, False)
        ctx.save_for_backward(buf, *ws)
        ctx.cfg = (cin0, E, has_base, tuple(widths), int(nrelu))
        ctx.wobjs, ctx.bobjs = ws, bs
        # third output: the slots that bound |buf| (parts + est), for a consumer of the whole buffer (the context network)
        slots = S.slots[S.first:S.first + S.n] if S is not None else None
        if slots is not None:
            ctx.mark_non_differentiable(slots)
        return buf, out, slots

    @staticmethod
    def backward(ctx, g_buf, g_out, _g_slots=None):
        buf = ctx.saved_tensors[0]
        ws = ctx.saved_tensors[1:]
        cin0, E, has_base, widths, nrelu = ctx.cfg
        nparts = len(widths)
        need_x = any(ctx.needs_input_grad[3:3 + nparts])
        B, _, H, W = buf.shape
        ctot = 448 + cin0
        dev = buf.device
        # G: gradient w.r.t. every channel of buf.  g_buf is produced exclusively for this node (the context
        # network's first conv), so it is updated in place.
        if g_buf is None:
            G = torch.zeros_like(buf)
        else:
            G = g_buf if (g_buf.is_contiguous() and g_buf.shape == buf.shape) else g_buf.contiguous()
        g_est = None
        if g_out is not None:
            g_est = g_out if _planes_dense(g_out) else g_out.contiguous()
        if has_base:
            g_est = add_planes(G[:, ctot:], g_est) if g_est is not None else G[:, ctot:].clone()
        grads_w = [None] * 6
        grads_b = [None] * 6
        # conv_last first: its data gradient touches every channel (K is tiny, the launch is memory-bound) and its
        # epilogue turns the c5 slice into a pre-activation gradient.  Then the buffer is back-propagated COLUMN-WISE:
        # for each slice T = c4, c3, c2, c1, x (in that order) ONE launch sums the contributions of all later layers,
        # reading their concatenated pre-activation gradients G[:, :t0] (contiguous by construction) against a
        # combined packed weight matrix, accumulates into G[:, T] once and applies LeakyReLU'(buf[:, T]) in the same
        # epilogue.  Versus layer-by-layer accumulation this replaces up to five small-K read-modify-write launches
        # per slice by a single large-K one.  Bias gradients ride on the wgrad launches.
        use_x3 = [(2 if (h2_code(B, t0, H, W, t1 - t0, 3, 1, 1)) else 1) if x3_code(B, t0, H, W, t1 - t0, 3, 1, 1) else 0
                  for (t0, t1) in ((32, 96), (96, 192), (192, 320), (320, 448), (448, ctot))]
        # fp16x2 route: one amax slot per slice of G, in channel order [g5, g4, g3, g2, g1, gx]; column k reads slices 0 .. k and
        # folds the magnitude of the slice it completes into slot k+1; slot 0 (the c5 slice) comes out of conv_last's data gradient
        S = ctx.amax
        Gs = Amax.zeros(dev, 6) if (S is not None or any(u == 2 for u in use_x3)) else None
        # one scale per channel for the weight gradients' gy-role operand (round 6): the column launches fold the channel maxima of the
        # gradient slice they complete (Gch); the layers whose launch runs with exchanged roles (Cout 96 / 32) need those of the
        # forward buffer instead: ONE pass over it serves both (Bch)
        ch_on = _c.WGRAD_CHANNEL_SCALE and S is not None and Gs is not None
        Gch = _c.zero_slots(dev, ctot) if ch_on else None
        Bch = [None]

        def chmax_for(t0_, t1_, cout_):
            """(x_chmax, gy_chmax) of the weight gradient of the layer with x = buf[:, t1_:ctot], gy = G[:, t0_:t1_]"""
            if not ch_on:
                return None, None
            if hip.lib().irr_conv2d_wgrad_h2_robust_side(B, ctot - t1_, H, W, cout_, 1):
                return None, (Gch[t0_:t1_] if Gch_valid[0] else None)
            if ctx.bch is not None:                          # folded slice by slice in the forward pass
                return ctx.bch[t1_:ctot], None
            if buf[:, :ctot].numel() * 4 > 4 * _c.WGRAD_CHANNEL_PASS_MAX_BYTES:
                return None, None

            def lazy():                                      # runs inside the weight-gradient launch (on the lane when there is one): the first
                if Bch[0] is None:                           # exchanged layer of the node measures the buffer, the second reuses it
                    Bch[0] = _c.channel_amax(buf[:, :ctot])
                return Bch[0][t1_:ctot]
            return lazy, None
        Gch_valid = [False]
        if g_est is not None:
            grads_w[5], grads_b[5] = wgrad_param(buf[:, :ctot], g_est, ctx.wobjs[5], ctx.bobjs[5], 1, 1)
            conv_dgrad(g_est, ws[5], 1, 1, (H, W), gx=G[:, :ctot], accumulate=True, mask=buf[:, :ctot], nmask=32,
                       gx_amax=Gs.sub(0) if (Gs is not None and ws[5].shape[0] <= 2) else None, amax_channels=32)
            if Gs is not None and ws[5].shape[0] > 2:
                amax_measure(G[:, :32], Gs.sub(0))
        else:
            lrelu_bwd_bias(G[:, :32], buf[:, :32], True, G[:, :32], None)
            if Gs is not None:
                amax_measure(G[:, :32], Gs.sub(0))
        packs = _dense_column_packs(ctx.wobjs[:5], cin0, tuple(use_x3))
        xc_, gc_ = chmax_for(0, 32, 32)                     # (conv5's gy, the c5 slice, comes from conv_last's data gradient: no fold there)
        grads_w[4], grads_b[4] = wgrad_param(buf[:, 32:ctot], G[:, :32], ctx.wobjs[4], ctx.bobjs[4], 1, 1,
                                             x_amax=S.sub(1, 5) if S is not None else None,
                                             gy_amax=Gs.sub(0) if Gs is not None else None, x_chmax=xc_, gy_chmax=gc_)   # conv5
        bounds = [(32, 96), (96, 192), (192, 320), (320, 448), (448, ctot)]
        for k_, (t0, t1) in enumerate(bounds):
            last = k_ == 4
            if last and not need_x:
                break
            # (the input column: its first `nrelu` channels are LeakyReLU outputs whose producer wants the PRE-activation gradient
            # -- the cost volume, models/IRR_PWC.py:94-95: the mask costs this MFMA-bound launch nothing, and the two HBM-bound
            # cost-volume gradient kernels no longer read their 81-plane output)
            nm = (nrelu if last else t1 - t0)
            margs = (hip.ptr(buf[:, t0:t1]), hip.bs(buf), nm) if nm > 0 else (None, 0, 0)
            LAUNCHES["dense_column_h2" if use_x3[k_] == 2 else "dense_column_x3" if use_x3[k_] else "dense_column_f32"] += 1
            if use_x3[k_]:
                args = ("irr_conv2d_fwd_x3", hip.ptr(G), hip.ptr(packs[k_]), None, None, hip.ptr(G[:, t0:t1]), B, t0, H, W,
                        t1 - t0, 1, hip.bs(G), hip.bs(G), 0, 0, 1.0, 1, *margs, hip.stream())
                code = x3_code(B, t0, H, W, t1 - t0, 3, 1, 1)
                variant = (_c.x3s_variant(use_x3[k_] == 2, None, True, nm > 0) if code == 9001
                           else (200000 if use_x3[k_] == 2 else 100000) + code)
                if use_x3[k_] == 2:
                    args, _ = _h2_args(args, G[:, :t0], Gs.sub(0, k_ + 1), Gs.sub(k_ + 1) if not last else None)
                    Gch_valid[0] = ch_on and not last and not (code == 9001 and _c._X3S_NO_CH_FOLD)      # (either kernel family folds them)
                    if Gch_valid[0]:
                        hip.lib().irr_conv_x3_next_chmax(Gch[t0:t1].data_ptr())      # (one-shot: this column's launch)
                else:
                    Gch_valid[0] = False
            else:
                Gch_valid[0] = False
                args = ("irr_conv2d_fwd_f32", hip.ptr(G), hip.ptr(packs[k_]), None, None, hip.ptr(G[:, t0:t1]), B, t0, H, W,
                        t1 - t0, H, W, 3, 1, 1, hip.bs(G), hip.bs(G), 0, 0, 1.0, 1, *margs, hip.stream())
                variant = hip.lib().irr_conv2d_fwd_variant(B, t1 - t0, H, W, 3)
            if _c.TIMER is None:
                _call_conv(args)
            else:
                _c.TIMER.wrap(variant, 2.0 * B * H * W * t0 * (t1 - t0) * 9, lambda: _call_conv(args), "dgrad",
                              nbytes=4.0 * B * H * W * (t0 + 2 * (t1 - t0) + nm))
            if not last:                                   # G[:, t0:t1] is now the pre-activation gradient of conv(4-k_)
                i = 3 - k_
                if Gs is not None and use_x3[k_] != 2:
                    amax_measure(G[:, t0:t1], Gs.sub(k_ + 1))
                xc_, gc_ = chmax_for(t0, t1, t1 - t0)
                grads_w[i], grads_b[i] = wgrad_param(buf[:, t1:ctot], G[:, t0:t1], ctx.wobjs[i], ctx.bobjs[i], 1, 1,
                                                     x_amax=S.sub(k_ + 2, 4 - k_) if S is not None else None,
                                                     gy_amax=Gs.sub(k_ + 1) if (Gs is not None and S is not None) else None,
                                                     x_chmax=xc_, gy_chmax=gc_)
        # g_est is still being read by the asynchronous wgrad lane (conv_last): autograd gets its own copy, because
        # the engine may accumulate the other gradients of `base` into the returned tensor IN PLACE
        gbase = g_est.clone() if (has_base and ctx.needs_input_grad[1]) else None
        out = [None, gbase, None]
        c0 = 448
        for i, wd in enumerate(widths):                       # per-part gradients = channel slices of G (plane-dense views)
            out.append(G[:, c0:c0 + wd] if (need_x and ctx.needs_input_grad[3 + i]) else None)
            c0 += wd
        for i in range(6):
            out += [grads_w[i], grads_b[i]]
        return tuple(out)



def dense_estimator(x, base, weights_and_biases, preact_grad_channels: int = 0):
    """(buf, out) -- see _DenseEstimatorFn.  x: the estimator's input, or a sequence of tensors whose channel concatenation
    it is.  weights_and_biases = [w1, b1, ..., w5, b5, w_last, b_last].
    preact_grad_channels = n: the first n input channels are LeakyReLU(0.1) outputs and the gradient returned for them is the
    PRE-activation gradient (multiplied by LeakyReLU' of the stored input) -- their producer must then not apply the derivative
    again (functional.cost_volume(..., grad_is_preactivation=True))."""
    parts = tuple(x) if isinstance(x, (list, tuple)) else (x,)
    if not all(p_.is_cuda for p_ in parts):
        raise RuntimeError("irr_amd conv runs on the HIP device only (no CPU fallback)")
    if preact_grad_channels and preact_grad_channels != int(parts[0].shape[1]):
        raise ValueError("preact_grad_channels must cover exactly the first input part")
    buf, out, slots = _DenseEstimatorFn.apply(len(parts), base, int(preact_grad_channels), *parts, *weights_and_biases)
    if slots is not None:                                   # (read by conv_chain when this very tensor object is its input)
        buf.__dict__["_irr_amax"] = Amax(slots, 0, 7 if base is not None else 6)
    return buf, out


# ----------------------------------------------------------------------------------------------
# autograd: a sequential chain of conv() blocks as ONE node
# ----------------------------------------------------------------------------------------------
class _ConvChainFn(hip.Function):
    """y = [res +] conv_n(... conv_1(x)) for the purely sequential sub-networks (ContextNetwork /
    OccContextNetwork, models/pwc_modules.py:210-243; the 7-conv stacks of RefineFlow / RefineOcc,
    models/irr_modules.py:71-79,115-123; the (stride-2, stride-1) pairs of FeatureExtractor,
    models/pwc_modules.py:91-96).

    Backward walks the chain with no elementwise pass over the activations: the data-gradient launch of
    layer i multiplies its result by LeakyReLU'(a_{i-1}) in its epilogue, so it directly yields the
    pre-activation gradient layer i-1 needs, and every bias gradient comes out of the wgrad launch."""

    @staticmethod
    def forward(ctx, x, res, cfg, x_amax, *wb):
        ws, bs = wb[0::2], wb[1::2]
        x = x if _planes_dense(x) else x.contiguous()
        acts = []
        cur = x
        n = len(ws)
        # fp16x2 route: in_amax[i] = amax slot of layer i's input where that layer runs on the h2 kernel -- the chain input is
        # measured, every other slot is filled by the producing layer's launch (y_amax)
        shapes = [tuple(x.shape)]
        for i in range(n):
            oh, ow = _c.out_hw(shapes[-1][2], shapes[-1][3], ws[i].shape[2], cfg[i][0], cfg[i][1])
            shapes.append((shapes[-1][0], ws[i].shape[0], oh, ow))
        h2_in = [bool(h2_code(shapes[i][0], shapes[i][1], shapes[i][2], shapes[i][3], ws[i].shape[0], ws[i].shape[2], cfg[i][0], cfg[i][1]))
                 for i in range(n)]
        slots = Amax.zeros(x.device, n) if any(h2_in) else None
        in_amax = [slots.sub(i) if h2_in[i] else None for i in range(n)]
        if h2_in[0]:
            if x_amax is not None:                             # the producer of x already knows its magnitude
                in_amax[0] = x_amax
            else:
                amax_measure(x, in_amax[0])
        xch = [None] * (n + 1)                                 # channel maxima of layer i's INPUT, where its weight gradient wants them
        for i in range(n):
            stride, dil, lrelu = cfg[i]
            last = i == n - 1
            ya = in_amax[i + 1] if not last else None
            if last and res is not None:
                res_c = res if _planes_dense(res) else res.contiguous()
                if lrelu:
                    a = conv_forward(cur, ws[i], bs[i], stride, dil, True, x_amax=in_amax[i])      # keep the activation for its mask
                    acts.append(a)
                    cur = torch.add(res_c, a)
                else:
                    cur = conv_forward(cur, ws[i], bs[i], stride, dil, False, res=res_c, x_amax=in_amax[i])
                    acts.append(None)
            else:
                ych = None
                if (not last and _c.WGRAD_CHANNEL_SCALE and _c.MATH == "h2" and h2_in[i + 1] and ctx.needs_input_grad[4 + 2 * (i + 1)]
                        and ws[i + 1].shape[2] == 3 and cfg[i + 1][0] == 1
                        and not hip.lib().irr_conv2d_wgrad_h2_robust_side(shapes[i + 1][0], shapes[i + 1][1], shapes[i + 1][2], shapes[i + 1][3],
                                                                         ws[i + 1].shape[0], cfg[i + 1][1])):
                    ych = _c.zero_slots(x.device, ws[i].shape[0])      # layer i + 1's weight gradient scales THIS output channel by channel
                cur = conv_forward(cur, ws[i], bs[i], stride, dil, lrelu, x_amax=in_amax[i], y_amax=ya, y_chmax=ych)
                acts.append(cur)
                xch[i + 1] = ych
        ctx.xch = xch
        ctx.in_amax = in_amax
        ctx.cfg = cfg
        ctx.has_res = res is not None
        ctx.weight_objs = ws
        ctx.bias_objs = bs
        ctx.save_for_backward(x, *[a for a in acts[:-1]], *( [acts[-1]] if cfg[-1][2] else [] ))
        return cur

    @staticmethod
    def backward(ctx, gy):
        cfg = ctx.cfg
        n = len(cfg)
        saved = ctx.saved_tensors
        x = saved[0]
        acts = list(saved[1:n])                               # a_0 .. a_{n-2}
        a_last = saved[n] if cfg[-1][2] else None
        ws = ctx.weight_objs
        gy = gy if _planes_dense(gy) else gy.contiguous()
        # (copy: gy may still be read by the asynchronous wgrad lane while autograd accumulates into gres in place)
        gres = gy.clone() if (ctx.has_res and ctx.needs_input_grad[1]) else None
        dev = gy.device
        g = gy
        grads = [None] * (2 * n)
        in_amax = ctx.in_amax
        # layer i wants the magnitude of ITS output gradient when its weight gradient (input slot known) or its data gradient runs on
        # the h2 kernels; the data-gradient launch of layer i+1 folds it into a slot, the chain's own output gradient is measured
        gshape = [None] * n
        hw_ = tuple(x.shape)
        for i in range(n):
            oh, ow = _c.out_hw(hw_[2], hw_[3], ws[i].shape[2], cfg[i][0], cfg[i][1])
            hw_ = (hw_[0], ws[i].shape[0], oh, ow)
            gshape[i] = hw_
        need_g = [in_amax[i] is not None or ((i > 0 or ctx.needs_input_grad[0]) and _dgrad_h2(gshape[i], ws[i], cfg[i][0], cfg[i][1]))
                  for i in range(n)]
        gslots = Amax.zeros(dev, n) if any(need_g) else None
        ga = gslots.sub(n - 1) if need_g[n - 1] else None
        if cfg[-1][2]:                                        # activation on the chain output: one pass on a small tensor,
            gpre = torch.empty_like(gy)                       # which also folds max |pre-activation gradient| into its slot
            lrelu_bwd_bias(gy, a_last, True, gpre, None, amax=ga)
            g = gpre
        elif ga is not None:
            amax_measure(g, ga)
        gch = None                                             # channel maxima of g when the launch that produced it folded them (round 6)
        for i in range(n - 1, -1, -1):
            stride, dil, _ = cfg[i]
            inp = acts[i - 1] if i > 0 else x
            grads[2 * i], grads[2 * i + 1] = wgrad_param(inp, g, ws[i], ctx.bias_objs[i], stride, dil, x_amax=in_amax[i], gy_amax=ga,
                                                         gy_chmax=gch, x_chmax=ctx.xch[i])
            gxa = gslots.sub(i - 1) if (i > 0 and need_g[i - 1]) else None
            gch = None
            if i > 0:
                prev_lrelu = cfg[i - 1][2]
                # layer i - 1's weight gradient takes this launch's output as its gy: fold its channel maxima here when that launch
                # will scale it channel by channel (fp16x2 route, gy in the kernel's gy role)
                if (_c.WGRAD_CHANNEL_SCALE and _c.MATH == "h2" and in_amax[i - 1] is not None and gxa is not None and ws[i - 1].shape[2] == 3
                        and cfg[i - 1][0] == 1 and hip.lib().irr_conv2d_wgrad_h2_robust_side(gshape[i - 1][0], ws[i - 1].shape[1], gshape[i - 1][2],
                                                                                         gshape[i - 1][3], ws[i - 1].shape[0], cfg[i - 1][1])):
                    gch = _c.zero_slots(dev, inp.shape[1])
                g = conv_dgrad(g, ws[i], stride, dil, inp.shape[2:], mask=inp if prev_lrelu else None,
                               nmask=inp.shape[1] if prev_lrelu else 0, gy_amax=ga, gx_amax=gxa, gx_chmax=gch)
            elif ctx.needs_input_grad[0]:
                g = conv_dgrad(g, ws[i], stride, dil, inp.shape[2:], gy_amax=ga)
            else:
                g = None
            ga = gxa
        return (g, gres, None, None) + tuple(grads)


def conv_chain(x, layers, res=None):
    """layers: sequence of modules exposing .weight, .bias, .stride, .dilation, .is_relu (modules.ConvBlock)."""
    if not x.is_cuda:
        raise RuntimeError("irr_amd conv runs on the HIP device only (no CPU fallback)")
    cfg = tuple((int(l.stride), int(l.dilation), bool(l.is_relu)) for l in layers)
    wb = []
    for l in layers:
        wb += [l.weight, l.bias]
    xa = x.__dict__.get("_irr_amax")
    if not (isinstance(xa, Amax) and xa.slots.device == x.device and x.dim() == 4):
        xa = None
    return _ConvChainFn.apply(x, res, cfg, xa, *wb)


# ----------------------------------------------------------------------------------------------
# autograd: OccUpsampleNetwork (models/irr_modules.py:30-56) as ONE node
# ----------------------------------------------------------------------------------------------

_KEEP_LOG = [] if os.environ.get("IRR_OCCUP_KEEP_LOG") else None      # diagnosis: (name, tensor) of the node's gradient maps, tools/lane_race_probe.py
_LANE_OCCUP_INLINE = os.environ.get("IRR_LANE_OCCUP_INLINE", "0") != "0"    # lane schedule (c), VERDICT r5 next #5 (profiles/r6_lane.txt): levels 5-6 keep their weight gradients in line
_LANE_HOLD_OCCUP = os.environ.get("IRR_LANE_HOLD_OCCUP", "0") != "0"      # lane schedule (a) of VERDICT r4 item 4, profiles/r5_lane_schedules.txt


class _OccUpsampleFn(hip.Function):
    """x_in -> init_conv -> 3 x [x += 0.1 * res_convs(x)] (shared weights) -> x_init + res_end_conv(x) -> out_convs + occ.

    The network runs on 32-channel maps at 1/2 and full resolution (41 % of all conv activation traffic,
    SURVEY.md Appendix A (iv)), so elementwise passes are expensive here.  The backward therefore uses the
    epilogue features of the MFMA data-gradient launch for every skip connection and activation:
    ``g_x = g_y + dgrad(...)`` (res), ``0.1 *`` (alpha), ``*= LeakyReLU'(t)`` (mask) and ``+=`` (accumulate);
    bias gradients come from the wgrad launches.

    The input arrives as its parts (nearest-x2 occlusion map first, then the guide tensors of models/IRR_PWC.py:166-167): they
    are copied straight into the channel slices of ONE buffer (no torch.cat of the 10-channel guide and again of the 11-channel
    input at full resolution).  When the bf16x3 streaming kernel accepts the problem with 16 input channels, that buffer gets
    16 channels (five of them zero) and init_conv runs there with zero-padded weights -- forward and data gradient: the 11 -> 32
    layer at 384x448 was the largest launch left on the fp32-MFMA kernels (1.35 ms at 52 TFLOP/s; the streaming kernel is bound by
    writing the 32-channel map).  The weight gradient reads the 11 real channels of the same buffer."""

    @staticmethod
    def forward(ctx, nparts, mul_const, *args):
        parts = args[:nparts]
        w_init, b_init, w_r0, b_r0, w_r1, b_r1, w_end, b_end, w_out, b_out = args[nparts:]
        occ_up = parts[0] if _planes_dense(parts[0]) else parts[0].contiguous()
        B, _, H, W = occ_up.shape
        widths = tuple(int(p_.shape[1]) for p_ in parts)
        cin = sum(widths)
        cpad = 16 if (cin < 16 and x3_code(B, 16, H, W, w_init.shape[0], 3, 1, 1)) else cin
        x_in = torch.empty(B, cpad, H, W, device=occ_up.device, dtype=torch.float32)
        # fp16x2 route: one amax slot per activation that a 32-channel conv (forward, or its weight gradient) reads -- slot 0: x_in
        # (folded by the copy that assembles it), 1 + i: x_i, 5 + i: t_(i+1); the launches that produce them fold their maxima in the epilogue
        S = Amax.zeros(x_in.device, 8) if h2_code(B, w_r0.shape[1], H, W, w_r0.shape[0], 3, 1, 1) else None
        sl = (lambda i: S.sub(i)) if S is not None else (lambda i: None)
        cat_channels_into(x_in, (occ_up,) + tuple(parts[1:]), zero_tail=cpad - cin, amax=sl(0))
        w_first = _padded_cin(w_init, cpad) if cpad > cin else w_init
        # LeakyReLU' masks as bits (round 5): x_init and t_1..t_3 are re-read by the backward's masked data gradients ONLY for the sign
        # of each element -- where producer and consumer both run on the fp16x2 streaming kernel the forward launch writes one bit per
        # element beside its output and the data gradient reads 1/32 of the bytes (bits[0]: x_init, bits[1 + i]: t_(i+1))
        nch = w_r0.shape[0]
        use_bits = (S is not None and _c.x3s_bits_ok(B, cpad, H, W, nch) and _c.x3s_bits_ok(B, nch, H, W, nch)
                    and w_r0.shape[1] == nch and tuple(w_r1.shape[:2]) == (nch, nch))
        bits = ([torch.empty(_c.x3s_mask_words(B, H, W), dtype=torch.int32, device=x_in.device) for _ in range(4)]
                if use_bits else [None] * 4)
        x_init = conv_forward(x_in, w_first, b_init, 1, 1, True, real_cin=cin, x_amax=sl(0), y_amax=sl(1), bits_out=bits[0])
        xs = [x_init]
        ts = []
        for i in range(3):
            t = conv_forward(xs[-1], w_r0, b_r0, 1, 1, True, x_amax=sl(1 + i), y_amax=sl(5 + i), bits_out=bits[1 + i])
            ts.append(t)
            xs.append(conv_forward(t, w_r1, b_r1, 1, 1, False, res=xs[-1], alpha=mul_const, x_amax=sl(5 + i), y_amax=sl(2 + i)))
        e, x2 = conv_forward_skip(xs[-1], w_end, b_end, True, x_init, x_amax=sl(4))
        o = conv_forward(x2, w_out, b_out, 1, 1, True)
        out = torch.add(o, occ_up)
        if S is not None and _c._CHECK_FINITE == "slots":        # diagnosis (NOTES C.5): fused magnitudes against a pass over the tensors
            for name, t, i in (("x_in", x_in, 0), ("x0", xs[0], 1), ("x1", xs[1], 2), ("x2", xs[2], 3), ("x3", xs[3], 4),
                               ("t1", ts[0], 5), ("t2", ts[1], 6), ("t3", ts[2], 7)):
                _c._FINITE_LOG.append((f"occ_upsample {tuple(t.shape)} {name}", torch.linalg.vector_norm(t, ord=float("inf")),
                                       S.slots[S.first + i].clone()))
        ctx.amax = S
        ctx.mul_const = mul_const
        ctx.widths = widths
        ctx.wobjs = (w_init, w_r0, w_r1, w_end, w_out)
        ctx.bobjs = (b_init, b_r0, b_r1, b_end, b_out)
        ctx.use_bits = use_bits
        ctx.save_for_backward(x_in, xs[0], xs[1], xs[2], xs[3], ts[0], ts[1], ts[2], e, x2, o, *(bits if use_bits else ()))
        return out

    @staticmethod
    def backward(ctx, g_out):
        # (IRR_LANE_HOLD_OCCUP: the node's weight-gradient launches reach the lane only when its data-gradient chain has been issued)
        held = _c.SIDE.hold() if (_LANE_HOLD_OCCUP and _c.SIDE is not None) else False
        here = _c.SIDE.here() if (_LANE_OCCUP_INLINE and _c.SIDE is not None and not held) else False
        try:
            return _OccUpsampleFn._backward(ctx, g_out)
        finally:
            if held:
                _c.SIDE.release()
            if here:
                _c.SIDE.there()

    @staticmethod
    def _backward(ctx, g_out):
        x_in, x0, x1, x2r, x3, t1, t2, t3, e, x2, o = ctx.saved_tensors[:11]
        bits = list(ctx.saved_tensors[11:15]) if ctx.use_bits else [None] * 4
        w_init, w_r0, w_r1, w_end, w_out = ctx.wobjs
        b_init, b_r0, b_r1, b_end, b_out = ctx.bobjs
        mc = ctx.mul_const
        widths = ctx.widths
        nparts = len(widths)
        cin = sum(widths)
        dev = g_out.device
        hw_ = x0.shape[2:]
        g_out = g_out if _planes_dense(g_out) else g_out.contiguous()
        z = lambda n: torch.zeros(n, device=dev, dtype=torch.float32)
        # fp16x2 route: slots of the forward (S: 0 x_in, 1 + i x_i, 5 + i t_(i+1)) and one per gradient map of the backward
        # (G: 0 gpre_e (measured), 1 the running g_x, 2 gpre_t; re-zeroed slots would cost launches, so every map gets its own)
        S = ctx.amax
        G = Amax.zeros(dev, 16) if S is not None else None
        sl = (lambda i: S.sub(i)) if S is not None else (lambda i: None)
        gl = (lambda i: G.sub(i)) if G is not None else (lambda i: None)
        _wg = wgrad_param
        if os.environ.get("IRR_OCCUP_NO_WGRAD_SLOTS"):           # (diagnosis switch: the node's weight gradients stay on bf16x3)
            wgrad_param_ = lambda *a, **k: _wg(*a, **{kk: vv for kk, vv in k.items() if kk not in ("x_amax", "gy_amax")})
        else:
            wgrad_param_ = _wg
        # out = occ_up + lrelu(conv_out(x2))
        gpre_o = torch.empty_like(g_out)
        gb_out = z(w_out.shape[0])
        lrelu_bwd_bias(g_out, o, True, gpre_o, gb_out)                       # 1-channel tensor
        gw_out, _ = wgrad_param_(x2, gpre_o, w_out, None, 1, 1, want_bias=False)
        # x2 = x_init + e, e = lrelu(conv_end(x3)): the gradient of x2 is needed raw (g_x2: the skip into x_init) and multiplied by
        # LeakyReLU'(e) (gpre_e: into res_end_conv).  Both come out of the out_convs data-gradient launch where its quad kernel
        # applies (one pass less over two 32-channel full-resolution maps); the bias gradient then rides on the wgrad launch.
        B_, _, H_, W_ = x2.shape
        dual = (w_out.shape[0] == 1 and W_ % 4 == 0 and not os.environ.get("IRR_OCCUP_NO_DUAL_DGRAD"))       # (A/B switch)
        # one scale per channel for the weight gradients' gy-role operand (round 6): the launches that PRODUCE the node's gradient maps fold
        # their channel maxima (the streaming kernel's epilogue waves keep one running maximum per channel anyway; the dual small-Cout
        # kernel reduces per wave and channel) -- no pass over the 0.35 / 1.4 GB maps of the 1/2- and full-resolution calls
        nch_ = w_r0.shape[0]
        ch_base = _c.WGRAD_CHANNEL_SCALE and G is not None and _c.MATH == "h2"

        def chv(cin_, cout_):
            """zeroed channel slots for the gy operand of a (cin_ -> cout_) layer's weight gradient at this map size, None when that
            launch does not scale its gy operand by channel"""
            if ch_base and bool(hip.lib().irr_conv2d_wgrad_h2_robust_side(B_, cin_, H_, W_, cout_, 1)):
                return _c.zero_slots(dev, cout_)
            return None
        if dual:
            g_x2 = torch.empty(B_, w_out.shape[1], H_, W_, device=dev, dtype=torch.float32)
            gpre_e = torch.empty_like(g_x2)
            ech = chv(w_end.shape[1], w_end.shape[0]) if not os.environ.get("IRR_DUAL_NO_CH_FOLD") else None      # (A/B)
            LAUNCHES["dgrad_smallco"] += 1
            hip.call("irr_conv2d_smallco_dgrad_dual_ch_f32", hip.ptr(gpre_o), hip.ptr(w_out.detach().contiguous()), hip.ptr(gpre_e),
                     hip.ptr(g_x2), hip.ptr(e), B_, w_out.shape[1], H_, W_, 1, hip.bs(gpre_o), hip.bs(gpre_e), hip.bs(g_x2), hip.bs(e),
                     G.sub(0).ptr() if G is not None else None, hip.ptr(ech), hip.stream())      # (max |gpre_e|, per tensor and per channel, folded by the same pass)
            gw_end, gb_end = wgrad_param_(x3, gpre_e, w_end, b_end, 1, 1, want_bias=True, x_amax=sl(4), gy_amax=gl(0), gy_chmax=ech)
        else:
            g_x2 = conv_dgrad(gpre_o, w_out, 1, 1, hw_)                      # (B,32,H,W); also the gradient of x_init via the skip
            gpre_e = torch.empty_like(g_x2)
            gb_end = z(w_end.shape[0])
            lrelu_bwd_bias(g_x2, e, True, gpre_e, gb_end)
            if G is not None:
                amax_measure(gpre_e, G.sub(0))
            gw_end, _ = wgrad_param_(x3, gpre_e, w_end, None, 1, 1, want_bias=False, x_amax=sl(4), gy_amax=gl(0))
        gch = chv(w_r1.shape[1], w_r1.shape[0])                # (g_x is the gy of conv_r1's weight gradient)
        g_x = conv_dgrad(gpre_e, w_end, 1, 1, hw_, gy_amax=gl(0), gx_amax=gl(1), gx_chmax=gch)     # gradient w.r.t. x3
        if _KEEP_LOG is not None:
            _KEEP_LOG.extend([(f"occ_upsample backward {tuple(g_out.shape)} g_out", g_out), (f"occ_upsample backward gpre_o", gpre_o),
                              (f"occ_upsample backward g_x2 (before the accumulate)", g_x2.clone())])
        def _slog(name, t, i):                                  # diagnosis (NOTES C.5), see forward
            if _KEEP_LOG is not None:                           # (NOTES D.5: references only, no extra launches; read after the pass)
                _KEEP_LOG.append((f"occ_upsample backward {tuple(t.shape)} {name}", t, (G.slots, G.first + i) if G is not None else None))
            if G is not None and _c._CHECK_FINITE == "slots":
                _c._FINITE_LOG.append((f"occ_upsample backward {tuple(t.shape)} {name}", torch.linalg.vector_norm(t, ord=float("inf")),
                                       G.slots[G.first + i].clone()))
        _slog("gpre_e", gpre_e, 0)
        _slog("g_x3", g_x, 1)
        # three residual blocks with shared weights: x_i = x_{i-1} + mc * conv_r1(t_i), t_i = lrelu(conv_r0(x_{i-1}))
        routed = _c.SIDE is not None and _c.SIDE.route(w_r0, b_r0) is not None
        acc_r0 = None if routed else (torch.zeros_like(w_r0), z(w_r0.shape[0]))
        acc_r1 = None if routed else (torch.zeros_like(w_r1), z(w_r1.shape[0]))
        xs = [x0, x1, x2r]
        ts = [t1, t2, t3]
        gxs = 1                                                 # slot of the running g_x
        for i in (2, 1, 0):
            wgrad_param_(ts[i], g_x, w_r1, b_r1, 1, 1, alpha=mc, acc=acc_r1, x_amax=sl(5 + i), gy_amax=gl(gxs), gy_chmax=gch)
            tch = chv(w_r0.shape[1], w_r0.shape[0])
            gpre_t = conv_dgrad(g_x, w_r1, 1, 1, hw_, mask=ts[i], nmask=ts[i].shape[1], alpha=mc, gy_amax=gl(gxs), gx_amax=gl(2 + 2 * i),
                                mask_bits=bits[1 + i], gx_chmax=tch)
            _slog(f"gpre_t{i}", gpre_t, 2 + 2 * i)
            wgrad_param_(xs[i], gpre_t, w_r0, b_r0, 1, 1, acc=acc_r0, x_amax=sl(1 + i), gy_amax=gl(2 + 2 * i), gy_chmax=tch)
            if i > 0:
                gch = chv(w_r1.shape[1], w_r1.shape[0])
                g_x = conv_dgrad(gpre_t, w_r0, 1, 1, hw_, res=g_x, gy_amax=gl(2 + 2 * i), gx_amax=gl(3 + 2 * i), gx_chmax=gch)      # skip + branch in one launch
                gxs = 3 + 2 * i
                _slog(f"g_x{i}", g_x, gxs)
            else:
                # x_0 = x_init: add the x2 skip gradient (accumulate into g_x2) and apply init_conv's LeakyReLU'
                ich = chv(cin, w_init.shape[0])                # (gpre_init is the gy of init_conv's weight gradient)
                conv_dgrad(gpre_t, w_r0, 1, 1, hw_, gx=g_x2, accumulate=True, res=g_x, mask=x0, nmask=x0.shape[1],
                           gy_amax=gl(2), gx_amax=gl(8), mask_bits=bits[0], gx_chmax=ich)
        gw_r0, gb_r0 = acc_r0 if acc_r0 is not None else (None, None)
        gw_r1, gb_r1 = acc_r1 if acc_r1 is not None else (None, None)
        _slog("gpre_init", g_x2, 8)
        gpre_init = g_x2
        x_real = x_in[:, :cin] if x_in.shape[1] > cin else x_in
        gw_init, gb_init = wgrad_param_(x_real, gpre_init, w_init, b_init, 1, 1, x_amax=sl(0), gy_amax=gl(8), gy_chmax=ich)
        gparts = [None] * nparts
        if any(ctx.needs_input_grad[2:2 + nparts]):
            w_first = _padded_cin(w_init, x_in.shape[1]) if x_in.shape[1] > cin else w_init
            g_xin = conv_dgrad(gpre_init, w_first, 1, 1, hw_, real_cin=cin, gy_amax=gl(8))
            if _KEEP_LOG is not None:
                _KEEP_LOG.append((f"occ_upsample backward {tuple(g_xin.shape)} g_xin", g_xin))
            c0 = 0
            for i, wd in enumerate(widths):
                if ctx.needs_input_grad[2 + i]:
                    gparts[i] = g_xin[:, c0:c0 + wd]
                c0 += wd
        if ctx.needs_input_grad[2]:                           # occ_up: channel 0 of the input AND the final skip
            gparts[0] = add_planes(g_out, gparts[0]) if gparts[0] is not None else g_out
        return (None, None, *gparts, gw_init, gb_init, gw_r0, gb_r0, gw_r1, gb_r1, gw_end, gb_end, gw_out, gb_out)


def occ_upsample_net(occ_up, guide, mod):
    """mod: modules.OccUpsampleNetwork.  occ_up = nearest-x2 occlusion map, guide = the guide tensor or the sequence of tensors
    whose channel concatenation it is; the network's input is cat([occ_up, guide])."""
    parts = (occ_up,) + (tuple(guide) if isinstance(guide, (list, tuple)) else (guide,))
    if not all(p_.is_cuda for p_ in parts):
        raise RuntimeError("irr_amd conv runs on the HIP device only (no CPU fallback)")
    return _OccUpsampleFn.apply(len(parts), float(mod.mul_const), *parts, mod.init_conv.weight, mod.init_conv.bias, mod.res_convs[0].weight,
                                mod.res_convs[0].bias, mod.res_convs[1].weight, mod.res_convs[1].bias,
                                mod.res_end_conv.weight, mod.res_end_conv.bias, mod.out_convs.weight, mod.out_convs.bias)
