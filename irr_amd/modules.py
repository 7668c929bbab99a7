"""Host-side building blocks of IRR-PWC, mirroring the names of the reference's
models/pwc_modules.py and models/irr_modules.py so that ``state_dict`` keys, constructor arguments and
call signatures are interchangeable, while every tensor operation runs in libirr_hip.so.

Parameter containers are real ``nn.Conv2d`` modules placed at the same positions as in the reference
(``conv(...)`` returns a Sequential whose element 0 is the Conv2d), so (a) the 124 state_dict keys match
and (b) ``torch.manual_seed(s); PWCNet(args)`` consumes the RNG in the same order and reproduces the
reference's MSRA initialisation bit for bit (tests/test_model_cpu.py).
"""
from __future__ import annotations

import logging

import torch
import torch.nn as nn

from . import conv as C
from . import functional as Fn


class ConvBlock(nn.Sequential):
    """conv(in_planes, out_planes, kernel_size, stride, dilation, isReLU) -- models/pwc_modules.py:8-19."""

    def __init__(self, in_planes, out_planes, kernel_size=3, stride=1, dilation=1, isReLU=True):
        layers = [nn.Conv2d(in_planes, out_planes, kernel_size=kernel_size, stride=stride, dilation=dilation,
                            padding=((kernel_size - 1) * dilation) // 2, bias=True)]
        if isReLU:
            layers.append(nn.LeakyReLU(0.1, inplace=True))
        super().__init__(*layers)
        self.stride, self.dilation, self.is_relu = stride, dilation, isReLU

    @property
    def weight(self):
        return self[0].weight

    @property
    def bias(self):
        return self[0].bias

    def forward(self, x, res=None, alpha=1.0):
        return C.conv_block(x, self[0].weight, self[0].bias, self.stride, self.dilation, self.is_relu, res, alpha)


def conv(in_planes, out_planes, kernel_size=3, stride=1, dilation=1, isReLU=True):
    return ConvBlock(in_planes, out_planes, kernel_size, stride, dilation, isReLU)


def initialize_msra(modules):
    """models/pwc_modules.py:22-39 (kaiming_normal_ weights, zero biases, in modules() order)."""
    logging.info("Initializing MSRA")
    for layer in modules:
        if isinstance(layer, (nn.Conv2d, nn.ConvTranspose2d)):
            nn.init.kaiming_normal_(layer.weight)
            if layer.bias is not None:
                nn.init.constant_(layer.bias, 0)


def upsample2d_as(inputs, target_as, mode="bilinear"):
    return Fn.upsample2d_as(inputs, target_as, mode)


def rescale_flow(flow, div_flow, width_im, height_im, to_local=True):
    """Scale (u, v) between full-resolution and level-local units (models/pwc_modules.py:70-82).

    NOTE: the reference mutates ``flow`` in place and returns a copy; this version is pure.  The one
    place where the mutation is observable (models/IRR_PWC.py:128-138) is written out explicitly in
    ``PWCNet.forward``."""
    if to_local:
        u_scale = float(flow.size(3) / width_im / div_flow)
        v_scale = float(flow.size(2) / height_im / div_flow)
    else:
        u_scale = float(width_im * div_flow / flow.size(3))
        v_scale = float(height_im * div_flow / flow.size(2))
    return flow * flow.new_tensor([u_scale, v_scale]).view(1, 2, 1, 1)


class FeatureExtractor(nn.Module):
    """models/pwc_modules.py:85-104."""

    def __init__(self, num_chs):
        super().__init__()
        self.num_chs = num_chs
        self.convs = nn.ModuleList()
        for ch_in, ch_out in zip(num_chs[:-1], num_chs[1:]):
            self.convs.append(nn.Sequential(conv(ch_in, ch_out, stride=2), conv(ch_out, ch_out)))

    def forward(self, x):
        pyramid = []
        for pair in self.convs:
            x = C.conv_chain(x, [pair[0], pair[1]])
            pyramid.append(x)
        return pyramid[::-1]


class WarpingLayer(nn.Module):
    """models/pwc_modules.py:115-133.  ``mask_threshold`` = 1.0 reproduces the reference's ``mask >= 1.0``."""

    def __init__(self, mask_threshold: float = 1.0):
        super().__init__()
        self.mask_threshold = mask_threshold

    def forward(self, x, flow, height_im, width_im, div_flow, swap_halves=False):
        return Fn.warp(x, flow, height_im, width_im, div_flow, self.mask_threshold, swap_halves)


class _DenseEstimator(nn.Module):
    def __init__(self, ch_in, ch_out):
        super().__init__()
        self.conv1 = conv(ch_in, 128)
        self.conv2 = conv(ch_in + 128, 128)
        self.conv3 = conv(ch_in + 256, 96)
        self.conv4 = conv(ch_in + 352, 64)
        self.conv5 = conv(ch_in + 416, 32)
        self.conv_last = conv(ch_in + 448, ch_out, isReLU=False)

    def _wb(self):
        out = []
        for layer in (self.conv1, self.conv2, self.conv3, self.conv4, self.conv5, self.conv_last):
            out += [layer.weight, layer.bias]
        return out

    def forward(self, x):
        """(x5, x_out) as in the reference; the five convs write into one buffer (new features PREPENDED)."""
        buf, out = C.dense_estimator(x, None, self._wb())
        return buf, out

    def forward_residual(self, x, base, preact_grad_channels: int = 0):
        """Model fast path: returns (cat([x5, est]), est) with est = base + conv_last(x5), i.e. the input of
        the context network (models/IRR_PWC.py:110-114) without building it by concatenation."""
        return C.dense_estimator(x, base, self._wb(), preact_grad_channels)


class FlowEstimatorDense(_DenseEstimator):
    """models/pwc_modules.py:153-170."""

    def __init__(self, ch_in):
        super().__init__(ch_in, 2)


class OccEstimatorDense(_DenseEstimator):
    """models/pwc_modules.py:190-207."""

    def __init__(self, ch_in):
        super().__init__(ch_in, 1)


class _Context(nn.Module):
    def __init__(self, ch_in, ch_out):
        super().__init__()
        self.convs = nn.Sequential(
            conv(ch_in, 128, 3, 1, 1), conv(128, 128, 3, 1, 2), conv(128, 128, 3, 1, 4), conv(128, 96, 3, 1, 8),
            conv(96, 64, 3, 1, 16), conv(64, 32, 3, 1, 1), conv(32, ch_out, isReLU=False))

    def forward(self, x, res=None):
        return C.conv_chain(x, list(self.convs), res=res)     # optional fused "est + context(...)"


class ContextNetwork(_Context):
    """models/pwc_modules.py:210-225."""

    def __init__(self, ch_in):
        super().__init__(ch_in, 2)


class OccContextNetwork(_Context):
    """models/pwc_modules.py:228-243."""

    def __init__(self, ch_in):
        super().__init__(ch_in, 1)


class OccUpsampleNetwork(nn.Module):
    """models/irr_modules.py:30-56."""

    def __init__(self, ch_in, ch_out):
        super().__init__()
        self.feat_dim = 32
        self.init_conv = conv(ch_in, self.feat_dim)
        self.res_convs = nn.Sequential(conv(self.feat_dim, self.feat_dim), conv(self.feat_dim, self.feat_dim, isReLU=False))
        self.res_end_conv = conv(self.feat_dim, self.feat_dim)
        self.mul_const = 0.1
        self.out_convs = conv(self.feat_dim, ch_out)

    def forward(self, occ, x):
        """x: the 10-channel guide, or the sequence of tensors whose channel concatenation it is"""
        occ = Fn.upsample_factor2(occ, x[0] if isinstance(x, (list, tuple)) else x)
        return C.occ_upsample_net(occ, x, self)


def subtract_mean(t):
    return t - t.mean(dim=(2, 3), keepdim=True)


def _refine_input(parts, first_conv):
    """cat(parts, dim=1) in one launch; when the first conv of the stack runs on the fp16x2 kernels the copy also folds the input's
    magnitude (no torch.cat + separate pass)"""
    B, _, H, W = parts[0].shape
    cin = sum(int(p_.shape[1]) for p_ in parts)
    w = first_conv.weight
    return C.cat_channels(parts, want_amax=bool(C.h2_code(B, cin, H, W, w.shape[0], w.shape[2], 1, 1)))


class _Refine(nn.Module):
    def __init__(self, ch_in):
        super().__init__()
        self.kernel_size = 3
        self.pad_size = 1
        self.convs = nn.Sequential(
            conv(ch_in, 128, 3, 1, 1), conv(128, 128, 3, 1, 1), conv(128, 64, 3, 1, 1), conv(64, 64, 3, 1, 1),
            conv(64, 32, 3, 1, 1), conv(32, 32, 3, 1, 1), conv(32, self.kernel_size * self.kernel_size, 3, 1, 1))


class RefineFlow(_Refine):
    """models/irr_modules.py:63-104.  ``scale`` folds the to_global rescale applied right after the call
    (models/IRR_PWC.py:137-138) into the tail kernel."""

    def forward(self, flow, diff_img, feature, scale=(1.0, 1.0)):
        flow_m = subtract_mean(flow)
        norm2_img = torch.linalg.vector_norm(diff_img, ord=2, dim=1, keepdim=True)
        feat = C.conv_chain(_refine_input((flow_m, norm2_img, feature), self.convs[0]), list(self.convs))
        return Fn.refine_tail(feat, flow, scale)


class RefineOcc(_Refine):
    """models/irr_modules.py:107-139."""

    def forward(self, occ, feat1, feat2):
        feat = C.conv_chain(_refine_input((occ, feat1, feat2), self.convs[0]), list(self.convs))
        return Fn.refine_tail(feat, occ, (1.0, 1.0))
