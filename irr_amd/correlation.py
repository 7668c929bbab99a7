"""``Correlation`` module with the operator signature of the reference's legacy CUDA package
(models/correlation_package/correlation.py:47-61):

    Correlation(pad_size, kernel_size, max_displacement, stride1, stride2, corr_multiply)(input1, input2)
        -> (B, ((max_displacement/stride2)*2+1)^2, outH, outW)

implemented by the MI355X cost-volume kernel (irr_corr81_{fwd,bwd}_f32) through a new-style static
autograd.Function (the reference's instance-style Function no longer runs on modern torch).
The hot path uses exactly one parameter point, (4, 1, 4, 1, 1, 1) (models/IRR_PWC.py:47); other values
are rejected with ValueError instead of being silently ignored as the reference's Python fallback does
(models/pwc_modules.py:42-53)."""
from __future__ import annotations

import torch.nn as nn

from . import functional as Fn


class Correlation(nn.Module):
    def __init__(self, pad_size=0, kernel_size=0, max_displacement=0, stride1=1, stride2=2, corr_multiply=1):
        super().__init__()
        self.pad_size = pad_size
        self.kernel_size = kernel_size
        self.max_displacement = max_displacement
        self.stride1 = stride1
        self.stride2 = stride2
        self.corr_multiply = corr_multiply
        if (pad_size, kernel_size, max_displacement, stride1, stride2) != (4, 1, 4, 1, 1):
            raise ValueError("Correlation: only (pad_size, kernel_size, max_displacement, stride1, stride2) = "
                             "(4, 1, 4, 1, 1) is implemented (the IRR-PWC operating point)")
        if corr_multiply != 1:
            raise ValueError("Correlation: only multiplicative correlation exists (as in the reference kernels)")

    def forward(self, input1, input2):
        return Fn.cost_volume(input1, input2, lrelu=False)
