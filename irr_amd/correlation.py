"""``Correlation`` module with the operator signature of the reference's legacy CUDA package
(models/correlation_package/correlation.py:47-61):

    Correlation(pad_size, kernel_size, max_displacement, stride1, stride2, corr_multiply)(input1, input2)
        -> (B, ((max_displacement/stride2)*2+1)^2, outH, outW)

through new-style static autograd Functions (the reference's instance-style Function no longer runs on modern torch).
The IRR-PWC operating point (4, 1, 4, 1, 1, 1) (models/IRR_PWC.py:47) runs on the tuned cost-volume kernels
(irr_corr81_{fwd,bwd}_f32); every other point -- none of the reference's models uses one -- on the general pair
irr_corr_general_{fwd,bwd}_f32 (round 5; forward arithmetic of correlation_cuda_kernel.cu:41-114, output shape of
correlation_cuda.cc:23-32, backward = the exact adjoint).  Rejected with ValueError: an even or non-positive kernel_size (the
reference's default 0 divides by zero, correlation_cuda_kernel.cu:74), corr_multiply != 1 (the reference kernels have no other
branch), a parameter point that leaves no output pixel."""
from __future__ import annotations

import ctypes

import torch
import torch.nn as nn

from . import functional as Fn
from . import hip


def output_shape(H: int, W: int, pad_size: int, kernel_size: int, max_displacement: int, stride1: int, stride2: int):
    """(channels, outH, outW) of the operator for H x W inputs (correlation_cuda.cc:23-32)"""
    c, oh, ow = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    rc = hip.lib().irr_corr_general_out_shape(H, W, pad_size, kernel_size, max_displacement, stride1, stride2,
                                              ctypes.byref(c), ctypes.byref(oh), ctypes.byref(ow))
    if rc != 0:
        raise ValueError(f"Correlation{(pad_size, kernel_size, max_displacement, stride1, stride2)} has no output for {H}x{W} inputs")
    return c.value, oh.value, ow.value


class _CorrelationGeneral(hip.Function):
    @staticmethod
    def forward(ctx, f1, f2, pad, k, md, s1, s2):
        if not (f1.is_cuda and f2.is_cuda):
            raise RuntimeError("irr_amd.Correlation runs on the HIP device only (no CPU fallback)")
        if f1.shape != f2.shape:
            raise ValueError(f"feature maps must have equal shapes, got {tuple(f1.shape)} vs {tuple(f2.shape)}")
        f1, f2 = f1.contiguous().float(), f2.contiguous().float()
        B, C, H, W = f1.shape
        ch, oh, ow = output_shape(H, W, pad, k, md, s1, s2)
        out = torch.empty(B, ch, oh, ow, device=f1.device, dtype=torch.float32)
        hip.call("irr_corr_general_fwd_f32", hip.ptr(f1), hip.ptr(f2), hip.ptr(out), B, C, H, W, pad, k, md, s1, s2,
                 hip.bs(f1), hip.bs(f2), hip.bs(out), hip.stream())
        ctx.save_for_backward(f1, f2)
        ctx.cfg = (pad, k, md, s1, s2)
        return out

    @staticmethod
    def backward(ctx, gout):
        f1, f2 = ctx.saved_tensors
        pad, k, md, s1, s2 = ctx.cfg
        gout = gout.contiguous()
        B, C, H, W = f1.shape
        g1 = torch.empty_like(f1) if ctx.needs_input_grad[0] else None
        g2 = torch.empty_like(f2) if ctx.needs_input_grad[1] else None
        hip.call("irr_corr_general_bwd_f32", hip.ptr(f1), hip.ptr(f2), hip.ptr(gout), hip.ptr(g1), hip.ptr(g2), B, C, H, W,
                 pad, k, md, s1, s2, hip.bs(f1), hip.bs(f2), hip.bs(gout), hip.bs(g1) if g1 is not None else 0,
                 hip.bs(g2) if g2 is not None else 0, hip.stream())
        return g1, g2, None, None, None, None, None


class Correlation(nn.Module):
    def __init__(self, pad_size=0, kernel_size=0, max_displacement=0, stride1=1, stride2=2, corr_multiply=1):
        super().__init__()
        self.pad_size = pad_size
        self.kernel_size = kernel_size
        self.max_displacement = max_displacement
        self.stride1 = stride1
        self.stride2 = stride2
        self.corr_multiply = corr_multiply
        if corr_multiply != 1:
            raise ValueError("Correlation: only multiplicative correlation exists (as in the reference kernels)")
        if kernel_size < 1 or kernel_size % 2 == 0:
            raise ValueError("Correlation: kernel_size must be a positive odd number (the reference's default 0 divides by zero, "
                             "correlation_cuda_kernel.cu:74)")
        if pad_size < 0 or max_displacement < 0 or stride1 < 1 or stride2 < 1:
            raise ValueError("Correlation: pad_size, max_displacement >= 0 and stride1, stride2 >= 1")

    def forward(self, input1, input2):
        pt = (self.pad_size, self.kernel_size, self.max_displacement, self.stride1, self.stride2)
        if pt == (4, 1, 4, 1, 1):
            return Fn.cost_volume(input1, input2, lrelu=False)          # the IRR-PWC point: the tuned kernels
        return _CorrelationGeneral.apply(input1, input2, *pt)
