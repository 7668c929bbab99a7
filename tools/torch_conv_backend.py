"""A/B tool (NOT part of the product): swaps irr_amd.conv's three primitives for torch's GPU convolution (MIOpen) so that the
HIP kernels can be timed / compared against the vendor library on the same shapes.

    import tools.torch_conv_backend as tb
    with tb.torch_convs():            # irr_amd.conv.conv_forward / conv_dgrad / conv_wgrad -> torch.nn.functional
        ...

The product module has no such switch: irr_amd.conv only ever launches libirr_hip.so."""
import contextlib

import torch
import torch.nn.functional as F

from irr_amd import conv as C


def conv_forward(x, weight, bias, stride, dil, lrelu, out=None, res=None, alpha=1.0, accumulate=False):
    k = weight.shape[2]
    v = F.conv2d(x, weight.detach(), bias.detach() if bias is not None else None, stride=stride,
                 padding=((k - 1) * dil) // 2, dilation=dil)
    if lrelu:
        v = F.leaky_relu(v, 0.1)
    v = v * alpha if res is None else res + alpha * v
    if out is None:
        return v
    if accumulate:
        out += v
    else:
        out.copy_(v)
    return out


def conv_dgrad(gy, weight, stride, dil, in_hw, gx=None, accumulate=False, mask=None, nmask=0, res=None, alpha=1.0):
    B = gy.shape[0]
    cin, k = weight.shape[1], weight.shape[2]
    v = torch.nn.grad.conv2d_input((B, cin) + tuple(in_hw), weight.detach(), gy, stride=stride,
                                   padding=((k - 1) * dil) // 2, dilation=dil)
    v = v * alpha if res is None else res + alpha * v
    if gx is None:
        gx = v
    elif accumulate:
        gx += v
    else:
        gx.copy_(v)
    if mask is not None and nmask > 0:
        gx[:, :nmask] *= torch.where(mask[:, :nmask] > 0, 1.0, 0.1)
    return gx


def conv_wgrad(x, gy, weight_shape, stride, dil, gw=None, gbias=None, alpha=1.0):
    cout, cin, k, _ = weight_shape
    if gw is None:
        gw = torch.zeros(cout, cin, k, k, device=x.device, dtype=torch.float32)
    gw += alpha * torch.nn.grad.conv2d_weight(x, (cout, cin, k, k), gy, stride=stride, padding=((k - 1) * dil) // 2, dilation=dil)
    if gbias is not None:
        gbias += alpha * gy.sum(dim=(0, 2, 3))
    return gw


@contextlib.contextmanager
def torch_convs():
    saved = (C.conv_forward, C.conv_dgrad, C.conv_wgrad)
    C.conv_forward, C.conv_dgrad, C.conv_wgrad = conv_forward, conv_dgrad, conv_wgrad
    try:
        yield
    finally:
        C.conv_forward, C.conv_dgrad, C.conv_wgrad = saved
