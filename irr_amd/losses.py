"""Multi-scale training loss of IRR-PWC -- drop-in for the reference's
``losses.MultiScaleEPE_PWC_Bi_Occ_upsample`` (losses.py:515-577): same constructor (``args`` with
``batch_size`` and ``model_div_flow``), same ``forward(output_dict, target_dict) -> loss_dict`` keys.

The one semantic change: the data-dependent Python branch ``if f_loss > o_loss`` (losses.py:560-567, a
device->host sync) is evaluated on the device with ``torch.where`` so the step never blocks and can be
captured in a hipGraph.  ``reduce_fn`` (optional) all-reduces the two detached scalars across data-parallel
ranks so the balancing weights equal those of a single-process run on the global batch.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as tf


def _elementwise_epe(input_flow, target_flow):
    return torch.linalg.vector_norm(target_flow - input_flow, ord=2, dim=1, keepdim=True)


def _downsample2d_as(inputs, target_as):
    h, w = target_as.shape[2:]
    return tf.adaptive_avg_pool2d(inputs, [h, w])


def f1_score_bal_loss(y_pred, y_true):
    """losses.py:39-48."""
    eps = 1e-8
    tp = -(y_true * torch.log(y_pred + eps)).sum(dim=(1, 2, 3))
    fn = -((1 - y_true) * torch.log((1 - y_pred) + eps)).sum(dim=(1, 2, 3))
    denom_tp = y_true.sum(dim=(1, 2, 3)) + y_pred.sum(dim=(1, 2, 3)) + eps
    denom_fn = (1 - y_true).sum(dim=(1, 2, 3)) + (1 - y_pred).sum(dim=(1, 2, 3)) + eps
    return ((tp / denom_tp).sum() + (fn / denom_fn).sum()) * y_pred.size(2) * y_pred.size(3) * 0.5


def fbeta_score(y_true, y_pred, beta, eps=1e-8):
    """losses.py:24-37."""
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
        self._weights = [0.32, 0.08, 0.02, 0.01, 0.005, 0.00125, 0.0003125]
        self.occ_activ = nn.Sigmoid()
        self.f1_score_bal_loss = f1_score_bal_loss
        self._reduce_fn = reduce_fn

    def forward(self, output_dict, target_dict):
        loss_dict = {}
        if self.training:
            output_flo, output_occ = output_dict['flow'], output_dict['occ']
            target_flo_f = self._args.model_div_flow * target_dict["target1"]
            target_flo_b = self._args.model_div_flow * target_dict["target2"]
            target_occ_f, target_occ_b = target_dict["target_occ1"], target_dict["target_occ2"]

            flow_loss = 0
            occ_loss = 0
            pooled = {}

            def pool(t, key, like):
                k = (key, like.shape[2], like.shape[3])
                if k not in pooled:
                    pooled[k] = _downsample2d_as(t, like)
                return pooled[k]

            for ii, output_ii in enumerate(output_flo):
                loss_ii = 0
                for jj in range(0, len(output_ii) // 2):
                    loss_ii = loss_ii + _elementwise_epe(output_ii[2 * jj], pool(target_flo_f, "ff", output_ii[2 * jj])).sum()
                    loss_ii = loss_ii + _elementwise_epe(output_ii[2 * jj + 1], pool(target_flo_b, "fb", output_ii[2 * jj + 1])).sum()
                flow_loss = flow_loss + self._weights[ii] * loss_ii / len(output_ii)

            for ii, output_ii in enumerate(output_occ):
                loss_ii = 0
                for jj in range(0, len(output_ii) // 2):
                    output_occ_f = self.occ_activ(output_ii[2 * jj])
                    output_occ_b = self.occ_activ(output_ii[2 * jj + 1])
                    loss_ii = loss_ii + self.f1_score_bal_loss(output_occ_f, pool(target_occ_f, "of", output_occ_f))
                    loss_ii = loss_ii + self.f1_score_bal_loss(output_occ_b, pool(target_occ_b, "ob", output_occ_b))
                occ_loss = occ_loss + self._weights[ii] * loss_ii / len(output_ii)

            f_loss, o_loss = flow_loss.detach(), occ_loss.detach()
            if self._reduce_fn is not None:
                f_loss, o_loss = self._reduce_fn(f_loss, o_loss)
            gt = f_loss > o_loss                          # losses.py:560-567, evaluated on the device
            one = torch.ones_like(f_loss)
            f_l_w = torch.where(gt, one, o_loss / f_loss)
            o_l_w = torch.where(gt, f_loss / o_loss, one)

            loss_dict["flow_loss"] = flow_loss / self._batch_size
            loss_dict["occ_loss"] = occ_loss / self._batch_size
            loss_dict["total_loss"] = (flow_loss * f_l_w + occ_loss * o_l_w) / self._batch_size
        else:
            loss_dict["epe"] = _elementwise_epe(output_dict["flow"], target_dict["target1"]).mean()
            loss_dict["F1"] = fbeta_score(target_dict["target_occ1"], torch.round(self.occ_activ(output_dict["occ"])), 1)
        return loss_dict
