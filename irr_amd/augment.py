"""On-GPU training augmentation, mirroring the reference's ``augmentations.RandomAffineFlowOcc``
(augmentations.py:368-653; selected by scripts/IRR-PWC_flyingChairsOcc.sh:34).

Split of work
  host    the random parameters: rejection-sampled affine thetas (augmentations.py:469-517), mirror signs (:66-96),
          noise stddev (:631) and crop origin (:565-585) -- a few dozen scalars per batch, drawn from torch's / numpy's
          global generators in the reference's order (so a CPU run of the reference with the same seeds yields the same
          parameters; tests/golden/augment.npz);
  device  everything per pixel, in two kernels of libirr_hip.so (csrc/augment.hip): ``irr_affine_warp_f32`` for the
          images and ``irr_affine_flow_occ_f32`` for flow + occlusion + out-of-bound test, both computing only the crop
          window.  There is no CPU fallback.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import hip

_IDENTITY = (1.0, 0.0, 0.0, 0.0, 1.0, 0.0)


def compose_thetas(theta0: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """apply_transform_to_params (augmentations.py:24-44): theta0 followed by the elementary transform t."""
    a, b = theta0.unbind(1), t.unbind(1)
    return torch.stack([a[0] * b[0] + a[3] * b[1], a[1] * b[0] + a[4] * b[1], b[2] + a[2] * b[0] + a[5] * b[1],
                        a[0] * b[3] + a[3] * b[4], a[1] * b[3] + a[4] * b[4], b[5] + a[2] * b[3] + a[5] * b[4]], dim=1)


def invert_thetas(thetas: torch.Tensor) -> torch.Tensor:
    """(B,6) -> (B,6) = (b1, b2, b4, b5, a3, a6): the linear part inverted, the translation kept
    (augmentations.py:421-433).  This is the ``inv`` argument of the kernels."""
    a = thetas.unbind(1)
    z = a[0] * a[4] - a[1] * a[3]
    return torch.stack([a[4] / z, -a[1] / z, -a[3] / z, a[0] / z, a[2], a[5]], dim=1)


def corners_leave_frame(thetas: torch.Tensor, height: int, width: int) -> torch.Tensor:
    """find_invalid (augmentations.py:439-466): True where one of the four image corners maps outside the frame."""
    inv = invert_thetas(thetas).unsqueeze(2)
    cx = thetas.new_tensor([-1.0, -1.0, 1.0, 1.0]) - inv[:, 4]
    cy = thetas.new_tensor([-1.0, 1.0, -1.0, 1.0]) - inv[:, 5]
    xq = 0.5 * (width - 1.0) * (inv[:, 0] * cx + inv[:, 1] * cy + 1.0)
    yq = 0.5 * (height - 1.0) * (inv[:, 2] * cx + inv[:, 3] * cy + 1.0)
    return ((xq < 0) | (yq < 0) | (xq >= width) | (yq >= height)).any(dim=1, keepdim=True)


def sample_thetas(theta0, max_translate, min_zoom, max_zoom, min_squeeze, max_squeeze, min_rotate, max_rotate, size):
    """apply_random_transforms_to_params (augmentations.py:468-517).  CPU tensors; global torch generator."""
    half_t = 0.5 * max_translate
    height, width = size
    n = theta0.size(0)
    draws = [theta0.new_zeros(n, 1) for _ in range(5)]
    ranges = [(min_zoom, max_zoom), (min_squeeze, max_squeeze), (-half_t, half_t), (-half_t, half_t), (min_rotate, max_rotate)]
    thetas = torch.zeros_like(theta0)
    todo = torch.ones(n, 1, dtype=torch.bool)
    while bool(todo.any()):
        for d, (lo, hi) in zip(draws, ranges):          # whole-batch redraw each round, as the reference does
            d.uniform_(lo, hi)
        zoom, squeeze, tx, ty, phi = draws
        sx, sy = zoom * squeeze, zoom / squeeze
        sin, cos = torch.sin(phi), torch.cos(phi)
        cand = compose_thetas(theta0, torch.cat([cos * sx, sin * sy, tx, -sin * sx, cos * sy, ty], dim=1))
        keep = todo.float()
        thetas = keep * cand + (1.0 - keep) * thetas
        todo = corners_leave_frame(thetas, height, width)
    return thetas


class RandomMirror(nn.Module):
    """augmentations.py:71-103: independent horizontal / vertical flips folded into both thetas."""

    def __init__(self, vertical=True, p=0.5):
        super().__init__()
        self._vertical, self._p = vertical, p

    def _signs(self, n):
        return torch.sign(2.0 * torch.bernoulli(torch.full((n, 1), float(self._p))) - 1.0)

    def forward(self, theta1, theta2):
        n = theta1.size(0)
        one = theta1.new_ones(n, 1)
        s = self._signs(n)
        m = torch.cat([s, s, s, one, one, one], dim=1)
        theta1, theta2 = theta1 * m, theta2 * m
        if self._vertical:
            s = self._signs(n)
            m = torch.cat([one, one, one, s, s, s], dim=1)
            theta1, theta2 = theta1 * m, theta2 * m
        return theta1, theta2


def _dev6(t: torch.Tensor, device) -> torch.Tensor:
    return t.to(device=device, dtype=torch.float32).contiguous()


def affine_warp(src, inv, window=None, noise=None, noise_std=0.0):
    """transform_image (augmentations.py:519-523) restricted to ``window`` = (y0, x0, OH, OW)."""
    if not src.is_cuda:
        raise hip.HipError("irr_amd.augment: CPU tensors are not supported (no fallback); move the batch to the GPU")
    src = src.contiguous()
    B, C, H, W = src.shape
    y0, x0, OH, OW = window if window is not None else (0, 0, H, W)
    dst = src.new_empty(B, C, OH, OW)
    with hip.device_of(src):
        hip.call("irr_affine_warp_f32", hip.ptr(src), hip.ptr(dst), hip.ptr(inv), hip.ptr(noise), float(noise_std), B, C, H, W, OH, OW,
                 y0, x0, C * H * W, C * OH * OW, hip.stream())
    return dst


def affine_flow_occ(flow, occ, inv_a, theta_a, theta_b, window=None):
    """transform_flow (:525-548) + transform_image(occ) + check_out_of_bound (:549-562) in one launch."""
    if not flow.is_cuda:
        raise hip.HipError("irr_amd.augment: CPU tensors are not supported (no fallback); move the batch to the GPU")
    flow = flow.contiguous()
    B, _, H, W = flow.shape
    y0, x0, OH, OW = window if window is not None else (0, 0, H, W)
    flow_out = flow.new_empty(B, 2, OH, OW)
    occ_out = None
    if occ is not None:
        occ = occ.contiguous()
        occ_out = occ.new_empty(B, 1, OH, OW)
    with hip.device_of(flow):
        hip.call("irr_affine_flow_occ_f32", hip.ptr(flow), hip.ptr(occ), hip.ptr(flow_out), hip.ptr(occ_out), hip.ptr(inv_a),
                 hip.ptr(theta_a), hip.ptr(theta_b), B, H, W, OH, OW, y0, x0,
                 2 * H * W, H * W, 2 * OH * OW, OH * OW, hip.stream())
    return flow_out, occ_out


class RandomAffineFlowOcc(nn.Module):
    """Same constructor and ``forward(example_dict)`` contract as augmentations.py:368-653."""

    def __init__(self, args, addnoise=True, crop=None):
        super().__init__()
        self._args = args
        self._addnoise = addnoise
        self._crop = crop
        self._mirror = RandomMirror(vertical=True, p=0.5)

    def sample(self, batch_size, height, width):
        """-> (theta1, theta2) on the CPU, after mirroring (augmentations.py:600-623)."""
        theta0 = torch.tensor([_IDENTITY], dtype=torch.float32).repeat(batch_size, 1)
        theta1 = sample_thetas(theta0, 0.2, 1.0, 1.5, 0.86, 1.16, -0.2, 0.2, [height, width])
        theta2 = sample_thetas(theta1, 0.015, 0.985, 1.015, 1.0, 1.0, -0.015, 0.015, [height, width])
        return self._mirror(theta1, theta2)

    def forward(self, example_dict, thetas=None):
        im1, im2 = example_dict["input1"], example_dict["input2"]
        B, _, H, W = im1.shape
        theta1, theta2 = self.sample(B, H, W) if thetas is None else thetas
        dev = im1.device
        inv1, inv2 = _dev6(invert_thetas(theta1.cpu()), dev), _dev6(invert_thetas(theta2.cpu()), dev)
        th1, th2 = _dev6(theta1, dev), _dev6(theta2, dev)

        std = float(np.random.uniform(0.0, 0.04)) if self._addnoise else 0.0       # augmentations.py:631
        window = None
        if self._crop is not None:                                                 # augmentations.py:565-585
            ch, cw = self._crop
            x0 = int(torch.empty(1, dtype=torch.int32).random_(0, W - cw + 1))
            y0 = int(torch.empty(1, dtype=torch.int32).random_(0, H - ch + 1))
            window = (y0, x0, ch, cw)
        OH, OW = (window[2], window[3]) if window else (H, W)

        def noise():
            return torch.randn(B, im1.size(1), OH, OW, device=dev, dtype=torch.float32) if self._addnoise else None

        out1 = affine_warp(im1, inv1, window, noise(), std)
        out2 = affine_warp(im2, inv2, window, noise(), std)
        flow_f, occ1 = affine_flow_occ(example_dict["target1"], example_dict["target_occ1"], inv1, th1, th2, window)
        flow_b, occ2 = affine_flow_occ(example_dict["target2"], example_dict["target_occ2"], inv2, th2, th1, window)

        example_dict["input1"], example_dict["input2"] = out1, out2
        example_dict["target1"], example_dict["target2"] = flow_f, flow_b
        example_dict["target_occ1"], example_dict["target_occ2"] = occ1, occ2
        return example_dict
