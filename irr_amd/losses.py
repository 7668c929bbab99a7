"""Multi-scale training loss of IRR-PWC -- drop-in for the reference's
``losses.MultiScaleEPE_PWC_Bi_Occ_upsample`` (losses.py:515-577): same constructor (``args`` with
``batch_size`` and ``model_div_flow``), same ``forward(output_dict, target_dict) -> loss_dict`` keys.

Everything that touches pixels runs in libirr_hip.so (target pyramid, EPE sums, balanced-F1 sums and their
gradients, csrc/loss.hip); what remains here is scalar algebra on a handful of device floats.

One semantic change: the data-dependent Python branch ``if f_loss > o_loss`` (losses.py:560-567, a
device->host sync) is evaluated on the device with ``torch.where`` so the step never blocks.  ``reduce_fn``
(optional) all-reduces the two detached scalars across data-parallel ranks so the balancing weights equal those
of a single-process run on the global batch.
"""
from __future__ import annotations

import ctypes
import os

import torch
import torch.nn as nn

from . import hip

_SINGLE_TERM_LAUNCHES = os.environ.get("IRR_LOSS_SINGLE") is not None
LEVEL_WEIGHTS = [0.32, 0.08, 0.02, 0.01, 0.005, 0.00125, 0.0003125]      # losses.py:522


def _need_cuda(*ts):
    for t in ts:
        if not t.is_cuda:
            raise RuntimeError("irr_amd loss kernels run on the HIP device only (no CPU fallback)")


def _dense(t):
    b, c, h, w = t.shape
    sb, sc, sh, sw = t.stride()
    ok = (sw == 1 or w == 1) and (sh == w or h == 1) and (sc == h * w or c == 1)
    return t if ok else t.contiguous()


def avg_pool_to(t: torch.Tensor, h: int, w: int, scale: float = 1.0) -> torch.Tensor:
    """scale * adaptive_avg_pool2d(t, [h, w]) (losses.py:16-18): s x s block means for the integer ratios of the /64 sizes,
    ATen's general windows otherwise (odd pyramid sizes)."""
    _need_cuda(t)
    B, C, H, W = t.shape
    t = t.contiguous()
    out = torch.empty(B, C, h, w, device=t.device, dtype=torch.float32)
    with hip.device_of(t):
        if H % h == 0 and W % w == 0 and H // h == W // w:
            hip.call("irr_avgpool_f32", hip.ptr(t), hip.ptr(out), B * C, h, w, H // h, float(scale), hip.stream())
        else:
            hip.call("irr_adaptive_avgpool_f32", hip.ptr(t), hip.ptr(out), B * C, H, W, h, w, float(scale), hip.stream())
    return out


class _EpeSum(hip.Function):
    """weight * sum_p ||tgt - flow||_2  (losses.py:8-10 with .sum())."""

    @staticmethod
    def forward(ctx, flow, tgt, weight: float):
        _need_cuda(flow, tgt)
        flow, tgt = _dense(flow), _dense(tgt)
        B, _, h, w = flow.shape
        out = torch.zeros(1, device=flow.device, dtype=torch.float32)
        part = _partials([flow], 1)
        hip.call("irr_epe_sum_fwd_f32", hip.ptr(flow), hip.ptr(tgt), hip.ptr(out), B, h * w, hip.bs(flow), hip.bs(tgt),
                 weight, hip.ptr(part), part.numel(), hip.stream())
        ctx.weight = weight
        ctx.save_for_backward(flow, tgt)
        return out

    @staticmethod
    def backward(ctx, g):
        flow, tgt = ctx.saved_tensors
        B, _, h, w = flow.shape
        g = g.contiguous()
        gf = torch.empty(B, 2, h, w, device=flow.device, dtype=torch.float32)
        hip.call("irr_epe_sum_bwd_f32", hip.ptr(flow), hip.ptr(tgt), hip.ptr(g), hip.ptr(gf), B, h * w, hip.bs(flow),
                 hip.bs(tgt), hip.bs(gf), ctx.weight, hip.stream())
        return gf, None, None


class _F1BalLoss(hip.Function):
    """weight * f1_score_bal_loss(sigmoid(logit), target)  (losses.py:39-48, 553-556)."""

    @staticmethod
    def forward(ctx, logit, tgt, weight: float):
        _need_cuda(logit, tgt)
        logit, tgt = _dense(logit), _dense(tgt)
        B, _, h, w = logit.shape
        sums = torch.empty(B, 4, device=logit.device, dtype=torch.float32)
        part = _partials([logit], 4)
        hip.call("irr_f1bal_sums_f32", hip.ptr(logit), hip.ptr(tgt), hip.ptr(sums), B, h * w, hip.bs(logit), hip.bs(tgt),
                 hip.ptr(part), part.numel(), hip.stream())
        out = torch.empty(1, device=logit.device, dtype=torch.float32)
        # (tp/(st+sp+eps)).sum() + (fn/((n-st)+(n-sp)+eps)).sum(), times h*w*0.5 * weight: one launch instead of 15 torch ops
        hip.call("irr_f1bal_value_f32", hip.ptr(sums), hip.ptr(out), B, h * w, float(weight * h * w * 0.5), hip.stream())
        ctx.weight = weight * h * w * 0.5
        ctx.save_for_backward(logit, tgt, sums)
        return out

    @staticmethod
    def backward(ctx, g):
        logit, tgt, sums = ctx.saved_tensors
        B, _, h, w = logit.shape
        g = g.contiguous()
        gl = torch.empty(B, 1, h, w, device=logit.device, dtype=torch.float32)
        hip.call("irr_f1bal_bwd_f32", hip.ptr(logit), hip.ptr(tgt), hip.ptr(sums), hip.ptr(g), hip.ptr(gl), B, h * w,
                 hip.bs(logit), hip.bs(tgt), hip.bs(gl), ctx.weight, hip.stream())
        return gl, None, None


class _Term(ctypes.Structure):
    """IrrLossTerm of include/irr_hip.h"""
    _fields_ = [("pred", ctypes.c_void_p), ("tgt", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("aux", ctypes.c_void_p),
                ("hw", ctypes.c_long), ("pred_bs", ctypes.c_long), ("tgt_bs", ctypes.c_long), ("grad_bs", ctypes.c_long),
                ("weight", ctypes.c_float), ("B", ctypes.c_int), ("nbx", ctypes.c_int), ("block0", ctypes.c_int)]


MAX_TERMS = 32          # IRR_LOSS_MAX_TERMS


def _partials(preds, per_block: int) -> torch.Tensor:
    """scratch for the forward reductions: one slot (EPE) / four slots (F1) per block.  The blocks store their partial sums
    there and one finishing block adds them in a fixed order (csrc/loss.hip): no atomics, bit-reproducible loss values."""
    lib = hip.lib()
    n = sum(int(lib.irr_loss_partial_blocks(p.shape[0], p.shape[2] * p.shape[3])) for p in preds)
    return torch.empty(per_block * n, device=preds[0].device, dtype=torch.float32)


def _paired_grads(preds):
    """gradient buffers for the terms' predictions.  The loss receives the two flow directions of an output as consecutive
    terms, both halves of one 2B-sample tensor of the model (irr_pwc._SplitHalves): their gradients are allocated as the two
    halves of ONE buffer, so the split's backward hands that buffer on instead of concatenating two tensors."""
    grads = []
    i = 0
    while i < len(preds):
        p = preds[i]
        if i + 1 < len(preds) and preds[i + 1].shape == p.shape:
            buf = torch.empty((2 * p.shape[0],) + tuple(p.shape[1:]), device=p.device, dtype=torch.float32)
            grads += [buf[:p.shape[0]], buf[p.shape[0]:]]
            i += 2
        else:
            grads.append(torch.empty(p.shape, device=p.device, dtype=torch.float32))
            i += 1
    return grads


def _term_table(preds, tgts, weights, grads=None, aux=None):
    n = len(preds)
    arr = (_Term * n)()
    for i in range(n):
        p, t = preds[i], tgts[i]
        e = arr[i]
        e.pred, e.tgt = p.data_ptr(), t.data_ptr()
        e.grad = grads[i].data_ptr() if grads is not None else None
        e.aux = aux[i].data_ptr() if aux is not None else None
        e.hw = p.shape[2] * p.shape[3]
        e.pred_bs, e.tgt_bs = hip.bs(p), hip.bs(t)
        e.grad_bs = hip.bs(grads[i]) if grads is not None else 0
        e.weight = weights[i]
        e.B = p.shape[0]
    return arr


class _MultiEpe(hip.Function):
    """sum_i weight_i * sum_p ||tgt_i - flow_i||_2 over ALL flow terms of the loss in one launch (and one for the gradients)."""

    @staticmethod
    def forward(ctx, weights, *tensors):
        n = len(weights)
        preds = [_dense(t) for t in tensors[:n]]
        tgts = [_dense(t) for t in tensors[n:]]
        _need_cuda(*preds, *tgts)
        out = torch.zeros(1, device=preds[0].device, dtype=torch.float32)
        for i0 in range(0, n, MAX_TERMS):
            sl = slice(i0, i0 + MAX_TERMS)
            arr = _term_table(preds[sl], tgts[sl], weights[sl])
            part = _partials(preds[sl], 1)
            hip.call("irr_epe_sum_multi_fwd_f32", ctypes.addressof(arr), len(arr), hip.ptr(out), hip.ptr(part), part.numel(),
                     hip.stream())
        ctx.weights = weights
        ctx.n = n
        ctx.save_for_backward(*preds, *tgts)
        return out

    @staticmethod
    def backward(ctx, g):
        n = ctx.n
        preds, tgts = ctx.saved_tensors[:n], ctx.saved_tensors[n:]
        g = g.contiguous()
        grads = _paired_grads(preds)
        for i0 in range(0, n, MAX_TERMS):
            sl = slice(i0, i0 + MAX_TERMS)
            arr = _term_table(preds[sl], tgts[sl], ctx.weights[sl], grads=grads[sl])
            hip.call("irr_epe_sum_multi_bwd_f32", ctypes.addressof(arr), len(arr), hip.ptr(g), hip.stream())
        return (None, *grads, *([None] * n))


class _MultiF1Bal(hip.Function):
    """sum_i weight_i * f1_score_bal_loss(sigmoid(logit_i), target_i) over ALL occlusion terms: two launches forward (per-sample
    sums, per-term algebra), one backward."""

    @staticmethod
    def forward(ctx, weights, *tensors):
        n = len(weights)
        preds = [_dense(t) for t in tensors[:n]]
        tgts = [_dense(t) for t in tensors[n:]]
        _need_cuda(*preds, *tgts)
        B = preds[0].shape[0]
        sums = torch.empty(n, B, 4, device=preds[0].device, dtype=torch.float32)
        out = torch.zeros(1, device=preds[0].device, dtype=torch.float32)
        scaled = tuple(w * p.shape[2] * p.shape[3] * 0.5 for w, p in zip(weights, preds))       # losses.py:553-556
        for i0 in range(0, n, MAX_TERMS):
            sl = slice(i0, i0 + MAX_TERMS)
            arr = _term_table(preds[sl], tgts[sl], scaled[sl], aux=[sums[i] for i in range(i0, min(n, i0 + MAX_TERMS))])
            part = _partials(preds[sl], 4)
            hip.call("irr_f1bal_multi_fwd_f32", ctypes.addressof(arr), len(arr), hip.ptr(out), hip.ptr(part), part.numel(),
                     hip.stream())
        ctx.scaled = scaled
        ctx.n = n
        ctx.save_for_backward(sums, *preds, *tgts)
        return out

    @staticmethod
    def backward(ctx, g):
        n = ctx.n
        sums = ctx.saved_tensors[0]
        preds, tgts = ctx.saved_tensors[1:1 + n], ctx.saved_tensors[1 + n:]
        g = g.contiguous()
        grads = _paired_grads(preds)
        for i0 in range(0, n, MAX_TERMS):
            sl = slice(i0, i0 + MAX_TERMS)
            arr = _term_table(preds[sl], tgts[sl], ctx.scaled[sl], grads=grads[sl],
                              aux=[sums[i] for i in range(i0, min(n, i0 + MAX_TERMS))])
            hip.call("irr_f1bal_multi_bwd_f32", ctypes.addressof(arr), len(arr), hip.ptr(g), hip.stream())
        return (None, *grads, *([None] * n))


def multi_epe(preds, tgts, weights):
    return _MultiEpe.apply(tuple(float(w) for w in weights), *preds, *tgts)


def multi_f1_bal(preds, tgts, weights):
    return _MultiF1Bal.apply(tuple(float(w) for w in weights), *preds, *tgts)


def balance_and_total(flow_loss, occ_loss, batch_size, reduce_fn=None):
    """losses.py:560-571 on device scalars: the smaller of the two terms is scaled up to the larger one."""
    f_loss, o_loss = flow_loss.detach(), occ_loss.detach()
    if reduce_fn is not None:
        f_loss, o_loss = reduce_fn(f_loss, o_loss)
    gt = f_loss > o_loss
    one = torch.ones_like(f_loss)
    f_l_w = torch.where(gt, one, o_loss / f_loss)
    o_l_w = torch.where(gt, f_loss / o_loss, one)
    return {"flow_loss": flow_loss / batch_size, "occ_loss": occ_loss / batch_size,
            "total_loss": (flow_loss * f_l_w + occ_loss * o_l_w) / batch_size}


def fbeta_score(y_true, y_pred, beta, eps=1e-8):
    """losses.py:24-37 (evaluation metric only)."""
    beta2 = beta ** 2
    y_pred, y_true = y_pred.float(), y_true.float()
    true_positive = (y_pred * y_true).sum(dim=2).sum(dim=2)
    precision = true_positive / (y_pred.sum(dim=2).sum(dim=2) + eps)
    recall = true_positive / (y_true.sum(dim=2).sum(dim=2) + eps)
    return torch.mean(precision * recall / (precision * beta2 + recall + eps) * (1 + beta2))


class MultiScaleEPE_PWC_Bi_Occ_upsample(nn.Module):
    def __init__(self, args, reduce_fn=None):
        super().__init__()
        self._args = args
        self._batch_size = args.batch_size
        self._weights = list(LEVEL_WEIGHTS)
        self.occ_activ = nn.Sigmoid()
        self._reduce_fn = reduce_fn

    def forward(self, output_dict, target_dict):
        loss_dict = {}
        if self.training:
            output_flo, output_occ = output_dict['flow'], output_dict['occ']
            div = float(self._args.model_div_flow)
            t_flo = (target_dict["target1"], target_dict["target2"])           # div_flow folded into the pooling kernel
            t_occ = (target_dict["target_occ1"], target_dict["target_occ2"])
            # target pyramid: every level is pooled from the next finer one that was already built (a mean of equal-sized
            # block means is the block mean), so the full-resolution targets are read once instead of once per level
            sizes = sorted({(o.shape[2], o.shape[3]) for lvl in list(output_flo) + list(output_occ) for o in lvl}, reverse=True)
            pooled = {}
            for kind, srcs, scale in (("f", t_flo, div), ("o", t_occ, 1.0)):
                for idx in (0, 1):
                    cur, cur_hw = srcs[idx], tuple(srcs[idx].shape[2:])
                    first = True
                    for hw_ in sizes:
                        if hw_ == cur_hw and first and scale == 1.0:
                            pooled[(kind, idx) + hw_] = cur
                            continue
                        if cur_hw[0] % hw_[0] or cur_hw[1] % hw_[1] or cur_hw[0] // hw_[0] != cur_hw[1] // hw_[1]:
                            cur, cur_hw, first = srcs[idx], tuple(srcs[idx].shape[2:]), True       # not nested: pool from the source
                        cur = avg_pool_to(cur, hw_[0], hw_[1], scale if first else 1.0)
                        cur_hw, first = hw_, False
                        pooled[(kind, idx) + hw_] = cur

            def pool(kind, idx, like):
                return pooled[(kind, idx, like.shape[2], like.shape[3])]

            f_pred, f_tgt, f_w, o_pred, o_tgt, o_w = [], [], [], [], [], []
            for ii, output_ii in enumerate(output_flo):
                wgt = self._weights[ii] / len(output_ii)
                for jj in range(len(output_ii) // 2):
                    for d in (0, 1):                                              # forward / backward direction
                        o = output_ii[2 * jj + d]
                        f_pred.append(o)
                        f_tgt.append(pool("f", d, o))
                        f_w.append(wgt)
            for ii, output_ii in enumerate(output_occ):
                wgt = self._weights[ii] / len(output_ii)
                for jj in range(len(output_ii) // 2):
                    for d in (0, 1):
                        o = output_ii[2 * jj + d]
                        o_pred.append(o)
                        o_tgt.append(pool("o", d, o))
                        o_w.append(wgt)
            # all 24 flow terms in one launch, all 24 occlusion terms in two (csrc/loss.hip, multi-term kernels)
            if _SINGLE_TERM_LAUNCHES:                                             # A/B switch (IRR_LOSS_SINGLE=1)
                flow_loss = torch.cat([_EpeSum.apply(p_, t_, w_) for p_, t_, w_ in zip(f_pred, f_tgt, f_w)]).sum()
                occ_loss = torch.cat([_F1BalLoss.apply(p_, t_, w_) for p_, t_, w_ in zip(o_pred, o_tgt, o_w)]).sum()
            else:
                flow_loss = multi_epe(f_pred, f_tgt, f_w).sum()
                occ_loss = multi_f1_bal(o_pred, o_tgt, o_w).sum()
            loss_dict = balance_and_total(flow_loss, occ_loss, self._batch_size, self._reduce_fn)
        else:
            tgt = target_dict["target1"]
            loss_dict["epe"] = torch.linalg.vector_norm(tgt - output_dict["flow"], ord=2, dim=1, keepdim=True).mean()
            loss_dict["F1"] = fbeta_score(target_dict["target_occ1"], torch.round(self.occ_activ(output_dict["occ"])), 1)
        return loss_dict


# ----------------------------------------------------------------------------------------------
# Losses of the PWC-Net ablation ladder and the fine-tuning stages.  The per-pixel reductions
# reuse the HIP kernels above; the Sintel / KITTI variants add a robust (Charbonnier-type) EPE and a BCE term that are
# written with device-side torch elementwise ops (fine-tuning losses, not on the measured path).
# ----------------------------------------------------------------------------------------------
PWC_LEVEL_WEIGHTS = [0.32, 0.08, 0.02, 0.01, 0.005]                      # losses.py:351


def _eval_epe(output_dict, target_dict):
    return torch.linalg.vector_norm(target_dict["target1"] - output_dict["flow"], ord=2, dim=1, keepdim=True).mean()


class _PoolCache:
    """avg-pooled targets per (tensor, level size): every level size is pooled once per step."""

    def __init__(self, div):
        self.div, self.store = div, {}

    def __call__(self, src, like, is_flow):
        key = (id(src), like.shape[2], like.shape[3])
        if key not in self.store:
            self.store[key] = avg_pool_to(src, like.shape[2], like.shape[3], self.div if is_flow else 1.0)
        return self.store[key]


class MultiScaleEPE_PWC(nn.Module):
    """losses.py:344-371 (models/pwcnet.py, pwcnet_irr.py)."""

    def __init__(self, args):
        super().__init__()
        self._args, self._batch_size, self._weights = args, args.batch_size, list(PWC_LEVEL_WEIGHTS)

    def forward(self, output_dict, target_dict):
        if not self.training:
            return {"epe": _eval_epe(output_dict, target_dict)}
        pool = _PoolCache(float(self._args.model_div_flow))
        terms = [_EpeSum.apply(o, pool(target_dict["target1"], o, True), self._weights[i]) for i, o in enumerate(output_dict['flow'])]
        return {"total_loss": torch.cat(terms).sum() / self._batch_size}


class MultiScaleEPE_PWC_Bi(nn.Module):
    """losses.py:374-402 (pwcnet_bi.py, pwcnet_irr_bi.py)."""

    def __init__(self, args):
        super().__init__()
        self._args, self._batch_size, self._weights = args, args.batch_size, list(PWC_LEVEL_WEIGHTS)

    def forward(self, output_dict, target_dict):
        if not self.training:
            return {"epe": _eval_epe(output_dict, target_dict)}
        pool = _PoolCache(float(self._args.model_div_flow))
        terms = []
        for i, (of, ob) in enumerate(output_dict['flow']):
            terms.append(_EpeSum.apply(of, pool(target_dict["target1"], of, True), self._weights[i]))
            terms.append(_EpeSum.apply(ob, pool(target_dict["target2"], ob, True), self._weights[i]))
        return {"total_loss": torch.cat(terms).sum() / (2 * self._batch_size)}


class MultiScaleEPE_PWC_Occ(nn.Module):
    """losses.py:405-454 (pwcnet_occ.py, pwcnet_irr_occ.py)."""

    def __init__(self, args, reduce_fn=None):
        super().__init__()
        self._args, self._batch_size, self._weights = args, args.batch_size, list(PWC_LEVEL_WEIGHTS)
        self.occ_activ = nn.Sigmoid()
        self._reduce_fn = reduce_fn

    def forward(self, output_dict, target_dict):
        if not self.training:
            return {"epe": _eval_epe(output_dict, target_dict),
                    "F1": fbeta_score(target_dict["target_occ1"], torch.round(self.occ_activ(output_dict["occ"])), 1)}
        pool = _PoolCache(float(self._args.model_div_flow))
        f_terms = [_EpeSum.apply(o, pool(target_dict["target1"], o, True), self._weights[i]) for i, o in enumerate(output_dict['flow'])]
        o_terms = [_F1BalLoss.apply(o, pool(target_dict["target_occ1"], o, False), self._weights[i]) for i, o in enumerate(output_dict['occ'])]
        return balance_and_total(torch.cat(f_terms).sum(), torch.cat(o_terms).sum(), self._batch_size, self._reduce_fn)


class MultiScaleEPE_PWC_Bi_Occ(nn.Module):
    """losses.py:457-512 (pwcnet_occ_bi.py, pwcnet_irr_occ_bi.py)."""

    def __init__(self, args, reduce_fn=None):
        super().__init__()
        self._args, self._batch_size, self._weights = args, args.batch_size, list(PWC_LEVEL_WEIGHTS)
        self.occ_activ = nn.Sigmoid()
        self._reduce_fn = reduce_fn

    def forward(self, output_dict, target_dict):
        if not self.training:
            return {"epe": _eval_epe(output_dict, target_dict),
                    "F1": fbeta_score(target_dict["target_occ1"], torch.round(self.occ_activ(output_dict["occ"])), 1)}
        pool = _PoolCache(float(self._args.model_div_flow))
        f_terms, o_terms = [], []
        for i, (of, ob) in enumerate(output_dict['flow']):
            f_terms.append(_EpeSum.apply(of, pool(target_dict["target1"], of, True), self._weights[i]))
            f_terms.append(_EpeSum.apply(ob, pool(target_dict["target2"], ob, True), self._weights[i]))
        for i, (of, ob) in enumerate(output_dict['occ']):
            o_terms.append(_F1BalLoss.apply(of, pool(target_dict["target_occ1"], of, False), self._weights[i]))
            o_terms.append(_F1BalLoss.apply(ob, pool(target_dict["target_occ2"], ob, False), self._weights[i]))
        return balance_and_total(torch.cat(f_terms).sum(), torch.cat(o_terms).sum(), 2 * self._batch_size, self._reduce_fn)


def _robust_epe_char(flow, tgt):
    """(||tgt - flow||_2 + 0.01) ** 0.4 per pixel (losses.py:12-14)."""
    return torch.pow(torch.linalg.vector_norm(tgt - flow, ord=2, dim=1, keepdim=True) + 0.01, 0.4)


class MultiScaleEPE_PWC_Bi_Occ_upsample_Sintel(nn.Module):
    """losses.py:579-638: Sintel fine-tuning -- only the forward direction is supervised (robust EPE + summed BCE on the
    occlusion probability); the backward-direction outputs take no gradient."""

    def __init__(self, args, reduce_fn=None):
        super().__init__()
        self._args, self._batch_size, self._weights = args, args.batch_size, list(LEVEL_WEIGHTS)
        self.occ_activ = nn.Sigmoid()
        self._reduce_fn = reduce_fn

    def forward(self, output_dict, target_dict):
        if not self.training:
            return {"epe": _eval_epe(output_dict, target_dict),
                    "F1": fbeta_score(target_dict["target_occ1"], torch.round(self.occ_activ(output_dict["occ"])), 1)}
        pool = _PoolCache(float(self._args.model_div_flow))
        flow_loss, occ_loss = 0, 0
        for ii, output_ii in enumerate(output_dict['flow']):
            loss_ii = 0
            for jj in range(len(output_ii) // 2):
                o = output_ii[2 * jj]
                loss_ii = loss_ii + _robust_epe_char(o, pool(target_dict["target1"], o, True)).sum()
            flow_loss = flow_loss + self._weights[ii] * loss_ii / len(output_ii) * 2
        for ii, output_ii in enumerate(output_dict['occ']):
            loss_ii = 0
            for jj in range(len(output_ii) // 2):
                o = output_ii[2 * jj]
                loss_ii = loss_ii + torch.nn.functional.binary_cross_entropy(self.occ_activ(o), pool(target_dict["target_occ1"], o, False),
                                                                             reduction='sum')
            occ_loss = occ_loss + self._weights[ii] * loss_ii / len(output_ii) * 2
        return balance_and_total(flow_loss, occ_loss, self._batch_size, self._reduce_fn)


class MultiScaleEPE_PWC_Bi_Occ_upsample_KITTI(nn.Module):
    """losses.py:640-699: KITTI fine-tuning -- sparse ground truth: every forward output is upsampled to full resolution,
    the robust EPE is masked by ``input_valid`` and normalised per sample by h*w / #valid; no occlusion term."""

    def __init__(self, args):
        super().__init__()
        self._args, self._batch_size = args, args.batch_size
        self._weights = [0.001, 0.001, 0.001, 0.002, 0.004, 0.004, 0.004]
        self.occ_activ = nn.Sigmoid()

    def forward(self, output_dict, target_dict):
        from . import functional as Fn
        valid = target_dict["input_valid"]
        b, _, h, w = target_dict["target1"].shape
        if not self.training:
            gt_mag = torch.linalg.vector_norm(target_dict["target1"], ord=2, dim=1, keepdim=True) + 1e-8
            epe = torch.linalg.vector_norm(target_dict["target1"] - output_dict["flow"], ord=2, dim=1, keepdim=True) * valid
            nvalid = valid.reshape(b, -1).sum(1)
            outlier = (epe > 3).float() * ((epe / gt_mag) > 0.05).float() * valid
            return {"epe": (epe.reshape(b, -1).sum(1) / nvalid).mean(), "outlier": (outlier.reshape(b, -1).sum(1) / nvalid).mean()}
        tgt = float(self._args.model_div_flow) * target_dict["target1"]
        norm_const = (h * w) / valid.reshape(b, -1).sum(1)                       # per sample
        flow_loss = 0
        for ii, output_ii in enumerate(output_dict['flow']):
            loss_ii = 0
            for jj in range(len(output_ii) // 2):
                up = Fn.resize_bilinear_ac(output_ii[2 * jj], h, w)
                e = _robust_epe_char(up, tgt) * valid
                loss_ii = loss_ii + (e.reshape(b, -1).sum(1) * norm_const).sum()
            flow_loss = flow_loss + self._weights[ii] * loss_ii / len(output_ii) * 2
        return {"flow_loss": flow_loss / self._batch_size, "total_loss": flow_loss / self._batch_size}
