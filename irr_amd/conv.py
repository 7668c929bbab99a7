"""conv() blocks of IRR-PWC on the MFMA kernels of libirr_hip.so: the fp32-faithful split-operand family (conv_x3 /
conv_x3s / conv_wgrad_x3, DESIGN.md 5.0: fp16x2 "h2" form by default, bf16x3 with ``IRR_CONV_MATH=x3``) wherever its
``*_eligible`` predicates accept the problem, the fp32-MFMA family elsewhere (``IRR_CONV_MATH=f32`` forces the latter everywhere).

Mirrors the reference helper ``conv(in_planes, out_planes, kernel_size, stride, dilation, isReLU)``
(models/pwc_modules.py:8-19, models/irr_modules.py:7-18): Conv2d with "same" padding and bias,
optionally followed by LeakyReLU(0.1).  All tensors may be channel-slice views of larger NCHW buffers
(dense H*W planes, arbitrary batch stride), which is how the DenseNet decoders avoid ``torch.cat``.

Layout (round 4): this module = the primitives (forward / data gradient / weight gradient launches), the kernel-family routing and
the run-time switches (``SIDE``, ``TIMER``, ``MATH``); ``conv_pack`` = packed-weight caches and the batched repack; ``conv_lane`` =
the weight-gradient lane and the batched folds; ``conv_nodes`` = the autograd nodes.  Everything is re-exported here.

There is no other backend in this module: every launch goes to libirr_hip.so (an A/B harness against torch's GPU
convolution lives in tools/torch_conv_backend.py, outside the product).
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch

from . import hip

from .conv_pack import (LAUNCHES, WEIGHT_EPOCH, _PackRegistry, _REGISTRIES, _announce_rewrite, _dense_column_packs, _packed,  # noqa: E402,F401
                        _padded_cin, _registry, _weight_tag, packed_weights)
from .conv_lane import ReduceBatch, WgradSide  # noqa: E402,F401

class KernelTimer:
    """Optional per-launch HIP-event timing of the MFMA conv kernel (bench.py's ``roofline`` object).
    Events are recorded on the launch stream right around the C-ABI call; nothing synchronises until
    ``summary()``.  Keyed by template instantiation (``conv_fwd_kernel<MT,NT,k>``)."""

    def __init__(self):
        self.records = []          # (variant code, flops, start event, stop event)
        self.active = True         # bench.py samples: events only on every N-th step of the timed region (an event pair costs
                                   # ~8 us of pipeline overlap per launch, 1.6 % of a step when recorded on every step)
        self.steps = 0             # steps on which events were recorded

    def begin_step(self, on: bool):
        self.active = bool(on)
        self.steps += 1 if on else 0

    def wrap(self, variant: int, flops: float, launch, phase: str = "fwd", nbytes: float = 0.0):
        """nbytes: ALGORITHMIC bytes of the launch (every operand map read once, every result map written once) -- what an
        HBM-bound kernel's roofline is priced with"""
        if not self.active:
            launch()
            return
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        launch()
        e.record()
        self.records.append((variant, flops, s, e, phase, nbytes))

    def summary(self):
        """{variant: {calls, flops, seconds, fwd_calls, fwd_flops, fwd_seconds}} -- the fwd_* entries cover only the
        forward-pass launches, which never share the chip with the asynchronous weight-gradient lane."""
        torch.cuda.synchronize()
        agg = {}
        for variant, flops, s, e, phase, nbytes in self.records:
            a = agg.setdefault(variant, [0, 0.0, 0.0, 0, 0.0, 0.0, 0.0, 0.0])
            dt = s.elapsed_time(e) * 1e-3
            a[0] += 1
            a[1] += flops
            a[2] += dt
            a[6] += nbytes
            if phase == "fwd":
                a[3] += 1
                a[4] += flops
                a[5] += dt
                a[7] += nbytes
        self.records.clear()
        return {v: {"calls": a[0], "flops": a[1], "seconds": a[2], "fwd_calls": a[3], "fwd_flops": a[4],
                    "fwd_seconds": a[5], "bytes": a[6], "fwd_bytes": a[7]} for v, a in agg.items()}


TIMER: Optional[KernelTimer] = None



def out_hw(h: int, w: int, k: int, stride: int, dil: int) -> Tuple[int, int]:
    pad = ((k - 1) * dil) // 2
    return ((h + 2 * pad - dil * (k - 1) - 1) // stride + 1, (w + 2 * pad - dil * (k - 1) - 1) // stride + 1)



# "h2" (default since round 4): fp16x2 of power-of-two-scaled operands, three MFMA products -- conv_x3_kernel<..., 2> /
# conv_wgrad_x3_kernel<..., true> wherever they are eligible and the bf16x3 streaming kernel for the 32-channel layers;
# "x3": bf16x3, six products, everywhere the x3 family is eligible (rounds 2-3 default); "f32": the fp32-MFMA kernels everywhere.
DEFAULT_MATH = os.environ.get("IRR_CONV_MATH", "h2")
MATH = DEFAULT_MATH


def set_math(name: str) -> None:
    """x3: bf16x3 (six products) wherever eligible; h2: fp16x2 of scaled operands (three products) wherever conv_x3_kernel /
    conv_wgrad_x3_kernel are eligible, bf16x3 on the streaming 32-channel kernel; f32: the fp32-MFMA kernels everywhere."""
    global MATH
    if name not in ("x3", "f32", "h2"):
        raise ValueError(name)
    MATH = name


_X3_ENV_DONE = [False]


def x3s_variant(h2: bool, res, accumulate: bool, masked: bool, dual: bool = False, bits: bool = False) -> int:
    """KernelTimer key of a conv_x3s_kernel<EPI, NP> launch: 1/2 00000 + 9010 + EPI (the launcher's choice, csrc/conv_x3.hip:
    4 = mask read as bits, 2 = accumulate / mask, 3 = second output, 1 = residual, 0 = plain) -- rocprofv3 lists the instantiations
    separately, so does the timer"""
    epi = 4 if bits else 2 if (accumulate or masked) else (3 if dual else 1) if res is not None else 0
    return (200000 if h2 else 100000) + 9010 + epi


def _map_bytes(B: int, H: int, W: int, *channels) -> float:
    return 4.0 * B * H * W * sum(channels)


def x3_code(B: int, cin: int, H: int, W: int, cout: int, k: int, stride: int, dil: int) -> int:
    if MATH not in ("x3", "h2"):
        return 0
    if not _X3_ENV_DONE[0]:
        _X3_ENV_DONE[0] = True
        if os.environ.get("IRR_X3_MIN_BLOCKS"):          # tests: exercise the kernel on small problems too
            hip.lib().irr_conv_x3_set_min_blocks(int(os.environ["IRR_X3_MIN_BLOCKS"]))
    return int(hip.lib().irr_conv2d_x3_eligible(B, cin, H, W, cout, k, stride, dil))



# The streaming 32-channel kernel (conv_x3s_kernel<EPI, NP>) in its fp16x2 form: ON by default since round 5 (+5 % on the step).
# Round 4 shipped it off: with the PLAIN low pieces of that round, `bench.py --gpus 2` on ONE shared GPU ended in a NaN loss in about
# every second run -- never in a single process.  Round 5 (profiles/NOTES.md C.5, profiles/r5_nan_ab.txt): same box, alternating
# libraries, the plain-low-piece build 7 of 8 runs NaN, the scaled-low-piece build (x3_split.h "Range") 0 of 31.
# IRR_X3S_H2=0 / set_x3s_h2(False): the bf16x3 form.
X3S_H2 = bool(int(os.environ.get("IRR_X3S_H2", "1")))
_X3S_NO_CH_FOLD = bool(os.environ.get("IRR_X3S_NO_CH_FOLD"))         # A/B: channel maxima of the streaming kernel's outputs by a pass (as before ABI 12)
_X3S_NO_FUSED_AMAX = bool(os.environ.get("IRR_X3S_NO_FUSED_AMAX"))     # diagnosis switch of NOTES C.5: the streaming kernel's output
                                                                       # magnitude by a separate pass instead of its epilogue


def set_x3s_h2(on: bool) -> bool:
    global X3S_H2
    old, X3S_H2 = X3S_H2, bool(on)
    return old


def h2_code(B: int, cin: int, H: int, W: int, cout: int, k: int, stride: int, dil: int) -> int:
    """non-zero: the problem runs on the fp16x2 form of conv_x3_kernel / conv_x3s_kernel (9001: the streaming one, see X3S_H2)"""
    if MATH != "h2" or not x3_code(B, cin, H, W, cout, k, stride, dil):
        return 0
    code = int(hip.lib().irr_conv2d_h2_eligible(B, cin, H, W, cout, k, stride, dil))
    return 0 if (code == 9001 and not X3S_H2) else code


def x3s_bits_ok(B: int, cin: int, H: int, W: int, cout: int) -> bool:
    """the 3x3 / stride-1 / dilation-1 layer runs on the fp16x2 streaming kernel with one co-tile: its LeakyReLU' mask can travel as
    bits (irr_conv2d_fwd_h2_bits; IRR_X3S_BITS=0: A/B switch, fp32 activations as masks)"""
    return X3S_BITS and cout <= 32 and h2_code(B, cin, H, W, cout, 3, 1, 1) == 9001


def x3s_mask_words(B: int, H: int, W: int) -> int:
    return int(hip.lib().irr_conv2d_x3s_mask_words(B, H, W))


X3S_BITS = os.environ.get("IRR_X3S_BITS", "1") != "0"
_X3S_BITS_NOREAD = bool(os.environ.get("IRR_X3S_BITS_NOREAD"))      # diagnosis: the forward writes the bits, the data gradients keep the fp32 masks

from .conv_pack import packed_weights_h2, packed_weights_x3  # noqa: E402,F401
from .conv_amax import Amax, measure as amax_measure  # noqa: E402,F401


# ----------------------------------------------------------------------------------------------
# primitives (no autograd)
# ----------------------------------------------------------------------------------------------
def _call_conv(args) -> None:
    """Launch a conv entry point from its argument tuple.  irr_conv2d_fwd_x3 problems that are too small to fill the chip
    get a scratch buffer and run through the K-split entry point (csrc/conv_x3.hip: blockIdx.z splits the channel chunks).
    A caller may have armed irr_conv_x3_next_chmax for this launch: if anything fails before the library consumes it (the scratch
    allocation), it is disarmed here -- a one-shot pointer must never reach a LATER launch of this thread."""
    try:
        _call_conv_(args)
    except BaseException:
        hip.lib().irr_conv_x3_next_chmax(None)
        raise


def _call_conv_(args) -> None:
    if args[0] in ("irr_conv2d_fwd_x3", "irr_conv2d_fwd_h2"):
        B, cin, H, W, cout, dil = args[6:12]
        n = hip.lib().irr_conv2d_fwd_x3_ws_elems(B, cin, H, W, cout, dil)
        ws = torch.empty(n, dtype=torch.float32, device=torch.device("cuda", torch.cuda.current_device())) if n > 0 else None
        if args[0] == "irr_conv2d_fwd_h2":                  # (args: the x3 tuple + (x_amax ptr, n_amax, y_amax ptr) before the stream)
            if n > 0 and KSPLIT_FUSED:
                # K-split launch (small pyramid levels): zeroed counters let the launch finish itself -- no finishing launch
                nc = int(hip.lib().irr_conv2d_fwd_x3_kcounters(B, cin, H, W, cout, dil))      # (0: a tile shape whose kernel has no in-launch finish)
                cnt = zero_slots(ws.device, max(nc, 1))
                hip.call("irr_conv2d_fwd_h2_kfused", *args[1:-4], hip.ptr(ws), n, hip.ptr(cnt), nc, *args[-4:])
                return
            hip.call("irr_conv2d_fwd_h2", *args[1:-4], hip.ptr(ws), n, *args[-4:])
            return
        if n > 0:
            hip.call("irr_conv2d_fwd_x3_splitk", *args[1:-1], ws.data_ptr(), n, args[-1])
            return
    hip.call(*args)


# IRR_X3_KSPLIT_FUSED=1: K-split launches of the fp16x2 kernel finish inside the launch (irr_conv2d_fwd_h2_kfused: the last block at a pixel
# tile sums the slices and runs the epilogue; 52 finishing launches per step less).  Bit-identical and, measured same-box, exactly as fast as
# the finishing launch (profiles/r6_ksplit_fused_ab.txt) -- OFF by default: it leans on agent-scope store / load ordering across the eight
# L2s where the two-launch route leans on a kernel boundary.
KSPLIT_FUSED = os.environ.get("IRR_X3_KSPLIT_FUSED", "0") != "0"

_CHECK_FINITE = os.environ.get("IRR_CONV_CHECK_FINITE", "")       # debugging aid: finiteness check behind every conv launch -- "1":
                                                                  # synchronising, raises at once; "async": device-side flags, read by
                                                                  # dump_finite_log() (irr_amd.train calls it when the loss is NaN)
_FINITE_LOG = []


def dump_finite_log(limit: int = 8) -> None:
    import sys
    torch.cuda.synchronize()
    if _CHECK_FINITE == "slots":                                   # entries: (what, measured max |t|, fused slot)
        bad = [(i, w, float(m), float(s)) for i, (w, m, s) in enumerate(_FINITE_LOG) if not float(s) >= float(m)]
        print(f"[slot log] {len(_FINITE_LOG)} tensors logged, {len(bad)} whose fused slot is below the measured magnitude (or not finite):", file=sys.stderr)
        for i, w, m, s in bad[:limit * 4]:
            print(f"  #{i} {w}: measured {m:.6e}, slot {s:.6e}", file=sys.stderr)
        return
    bad = [(i, e) for i, e in enumerate(_FINITE_LOG) if not bool(e[1])]
    print(f"[finite log] {len(_FINITE_LOG)} conv launches logged, {len(bad)} with a non-finite output; first:", file=sys.stderr)
    for i, (what, flag, slots, stats) in bad[:limit]:
        print(f"  #{i} {what}; amax slots {[None if s is None else s.tolist() for s in slots]}; max|out| / non-finite count {stats.tolist()}", file=sys.stderr)
        if i > 0:
            w0, f0, s0, st0 = _FINITE_LOG[i - 1]
            print(f"     previous launch #{i - 1} {w0}: finite {bool(f0)}, slots {[None if s is None else s.tolist() for s in s0]}, max|out| {st0.tolist()}", file=sys.stderr)


def _check_finite(what, out, *slots):
    if _CHECK_FINITE == "async":
        fin = torch.isfinite(out)
        stats = torch.stack([torch.where(fin, out, torch.zeros_like(out)).abs().max(), (~fin).sum().float()])
        _FINITE_LOG.append((what, fin.all(), [None if a is None else a.slots[a.first:a.first + a.n].clone() for a in slots], stats))
        if len(_FINITE_LOG) > 20000:
            del _FINITE_LOG[:10000]
        return
    if not torch.isfinite(out).all():
        vals = [None if a is None else a.slots[a.first:a.first + a.n].tolist() for a in slots]
        raise FloatingPointError(f"{what}: non-finite output {tuple(out.shape)}, amax slots {vals}, "
                                 f"non-finite {int((~torch.isfinite(out)).sum())} of {out.numel()}")


def _h2_args(x3_args, x, x_amax, y_amax):
    """the argument tuple of irr_conv2d_fwd_x3 turned into irr_conv2d_fwd_h2's (_call_conv adds the scratch): x's magnitude is
    measured here when the caller has no slot for it"""
    xa = x_amax if x_amax is not None else amax_measure(x)
    return ("irr_conv2d_fwd_h2",) + tuple(x3_args[1:-1]) + (xa.ptr(), xa.n, y_amax.ptr() if y_amax is not None else None, x3_args[-1]), xa


def conv_forward(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], stride: int, dil: int,
                 lrelu: bool, out: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None,
                 alpha: float = 1.0, accumulate: bool = False, real_cin: Optional[int] = None,
                 x_amax: Optional[Amax] = None, y_amax: Optional[Amax] = None, bits_out: Optional[torch.Tensor] = None,
                 y_chmax: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = [res +] alpha * act(conv(x, weight) + bias)   (accumulate: out += ...).
    real_cin: the layer's true input-channel count when x / weight are zero-padded copies (KernelTimer prices algorithmic FLOPs).
    x_amax: slots that bound |x| (MATH == "h2"; measured here when absent).  y_amax: a zeroed slot that holds max |out| after the
    call whatever kernel family ran (the h2 launch folds it in its epilogue, any other route costs one pass over out).
    bits_out: int32 tensor of x3s_mask_words(B, H, W) words that receives (out > 0) per element for conv_dgrad(mask_bits=...) -- the
    caller has checked x3s_bits_ok for this layer.
    y_chmax: a ZEROED (Cout,) tensor that holds max |out[:, c]| per channel after the call (as gx_chmax of conv_dgrad): the scales of a
    weight gradient that takes ``out`` as the operand in its kernel's gy role (a layer whose launch runs with exchanged roles)."""
    # (the LeakyReLU'-mask epilogue of the kernel is only used by conv_dgrad)
    B, cin, H, W = x.shape
    cout, cin_w, k, _ = weight.shape
    assert cin == cin_w, (x.shape, weight.shape)
    oh, ow = out_hw(H, W, k, stride, dil)
    if out is None:
        out = torch.empty(B, cout, oh, ow, device=x.device, dtype=torch.float32)
    assert out.shape == (B, cout, oh, ow), (out.shape, (B, cout, oh, ow))
    if cout <= 4 and stride == 1:
        wc = weight.detach().contiguous()
        LAUNCHES["fwd_smallco"] += 1
        hip.call("irr_conv2d_smallco_fwd_f32", hip.ptr(x), hip.ptr(wc), hip.ptr(bias.detach() if bias is not None else None),
                 hip.ptr(res), hip.ptr(out), B, cin, H, W, cout, k, dil, hip.bs(x), hip.bs(out),
                 hip.bs(res) if res is not None else 0, int(lrelu), float(alpha), int(accumulate), hip.stream())
        if y_amax is not None:
            amax_measure(out, y_amax)
        return out
    code = x3_code(B, cin, H, W, cout, k, stride, dil)
    h2 = bool(code) and bool(h2_code(B, cin, H, W, cout, k, stride, dil))
    if code:
        wq = packed_weights_h2(weight, False) if h2 else packed_weights_x3(weight, False)
        args = ("irr_conv2d_fwd_x3", hip.ptr(x), hip.ptr(wq), hip.ptr(bias.detach() if bias is not None else None),
                hip.ptr(res), hip.ptr(out), B, cin, H, W, cout, dil,
                hip.bs(x), hip.bs(out), hip.bs(res) if res is not None else 0,
                int(lrelu), float(alpha), int(accumulate), None, 0, 0, hip.stream())
        measure_y = h2 and code == 9001 and y_amax is not None and _X3S_NO_FUSED_AMAX
        if h2:
            args, _xa = _h2_args(args, x, x_amax, None if measure_y else y_amax)
        if bits_out is not None:
            if not (h2 and code == 9001 and res is None and not accumulate):
                raise ValueError("bits_out: only the plain forward of the fp16x2 streaming kernel writes bit masks")
            # (irr_conv2d_fwd_h2's tuple: ..., accumulate, mask, mask_bs, nmask, x_amax, n, y_amax, stream -> ..., accumulate, mask_bits, nmask, bits_out, ...)
            args = ("irr_conv2d_fwd_h2_bits",) + tuple(args[1:-7]) + (None, 0, bits_out.data_ptr()) + tuple(args[-4:])
        variant = x3s_variant(h2, res, accumulate, False) if code == 9001 else (200000 if h2 else 100000) + code
        LAUNCHES["fwd_x3s" if code == 9001 else "fwd_h2" if h2 else "fwd_x3"] += 1          # (fwd_x3s: the streaming kernel, either form)
    else:
        LAUNCHES["fwd_f32"] += 1
        wp = packed_weights(weight, False)
        args = ("irr_conv2d_fwd_f32", hip.ptr(x), hip.ptr(wp), hip.ptr(bias.detach() if bias is not None else None),
                hip.ptr(res), hip.ptr(out), B, cin, H, W, cout, oh, ow, k, stride, dil,
                hip.bs(x), hip.bs(out), hip.bs(res) if res is not None else 0,
                int(lrelu), float(alpha), int(accumulate), None, 0, 0, hip.stream())
        variant = None
    if y_chmax is not None and h2 and args[0] in ("irr_conv2d_fwd_h2", "irr_conv2d_fwd_h2_bits") and not (code == 9001 and (_X3S_NO_FUSED_AMAX or _X3S_NO_CH_FOLD)):
        hip.lib().irr_conv_x3_next_chmax(y_chmax.data_ptr())      # (one-shot: the launch below)
        y_chmax = None
    if TIMER is None:
        _call_conv(args)
    else:
        if variant is None:
            variant = hip.lib().irr_conv2d_fwd_variant(B, cout, oh, ow, k)
        TIMER.wrap(variant, 2.0 * B * oh * ow * cout * (real_cin or cin) * k * k, lambda: _call_conv(args),
                   nbytes=_map_bytes(B, H, W, cin) + _map_bytes(B, oh, ow, cout * (1 + (res is not None) + bool(accumulate))))
    if y_amax is not None and (not h2 or (code == 9001 and _X3S_NO_FUSED_AMAX)):
        amax_measure(out, y_amax)
    if y_chmax is not None:                                 # (no launch folded them)
        channel_amax(out, y_chmax)
    if _CHECK_FINITE and _CHECK_FINITE != "slots":
        _check_finite(f"conv_forward {tuple(x.shape)} -> {cout} code {code} h2 {h2}", out, x_amax, y_amax)
    return out


def conv_forward_skip(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], lrelu: bool, skip: torch.Tensor,
                      x_amax: Optional[Amax] = None, y_amax: Optional[Amax] = None, y_chmax: Optional[torch.Tensor] = None):
    """(e, y) with e = act(conv3x3(x) + bias) and y = skip + e.  On the streaming 32-channel kernel both come out of ONE launch
    (irr_conv2d_fwd_x3_dual / _h2_dual); elsewhere e is computed and the sum is an elementwise pass.  x_amax / y_amax: as in
    conv_forward (y_amax bounds y)."""
    B, cin, H, W = x.shape
    cout = weight.shape[0]
    if x3_code(B, cin, H, W, cout, 3, 1, 1) == 9001 and _planes_dense(skip) and not os.environ.get("IRR_X3S_NO_DUAL"):      # (A/B switch)
        h2 = bool(h2_code(B, cin, H, W, cout, 3, 1, 1))
        e = torch.empty(B, cout, H, W, device=x.device, dtype=torch.float32)
        y = torch.empty_like(e)
        wq = packed_weights_h2(weight, False) if h2 else packed_weights_x3(weight, False)
        LAUNCHES["fwd_x3s"] += 1
        args = ("irr_conv2d_fwd_h2_dual" if h2 else "irr_conv2d_fwd_x3_dual", hip.ptr(x), hip.ptr(wq),
                hip.ptr(bias.detach() if bias is not None else None), hip.ptr(skip),
                hip.ptr(y), hip.ptr(e), B, cin, H, W, cout, 1, hip.bs(x), hip.bs(y), hip.bs(skip), hip.bs(e), int(lrelu), 1.0)
        if h2:
            xa = x_amax if x_amax is not None else amax_measure(x)
            args += (xa.ptr(), xa.n, y_amax.ptr() if y_amax is not None else None)
        args += (hip.stream(),)
        if y_chmax is not None and h2 and not _X3S_NO_CH_FOLD:
            hip.lib().irr_conv_x3_next_chmax(y_chmax.data_ptr())      # (one-shot: the launch below; y_chmax bounds y, the sum)
            y_chmax = None
        if TIMER is None:
            hip.call(*args)
        else:
            TIMER.wrap(x3s_variant(h2, skip, False, False, dual=True), 2.0 * B * H * W * cout * cin * 9, lambda: hip.call(*args),
                       nbytes=_map_bytes(B, H, W, cin, 3 * cout))
        if y_amax is not None and not h2:
            amax_measure(y, y_amax)
        if y_chmax is not None:
            channel_amax(y, y_chmax)
        return e, y
    e = conv_forward(x, weight, bias, 1, 1, lrelu, x_amax=x_amax)
    y = torch.add(skip, e)
    if y_amax is not None:
        amax_measure(y, y_amax)
    if y_chmax is not None:
        channel_amax(y, y_chmax)
    return e, y


S2_GATHER_MAX_CIN = 96   # stride-2 3x3 data gradients with at most this many result channels use the 2x2-block kernel
                         # (measured at the BASELINE batch: faster up to 128 -> 96 at 24x28, slower for 196 -> 128 at 12x14; A/B: 0)


def conv_dgrad(gy: torch.Tensor, weight: torch.Tensor, stride: int, dil: int, in_hw: Tuple[int, int],
               gx: Optional[torch.Tensor] = None, accumulate: bool = False,
               mask: Optional[torch.Tensor] = None, nmask: int = 0,
               res: Optional[torch.Tensor] = None, alpha: float = 1.0, real_cin: Optional[int] = None,
               gy_amax: Optional[Amax] = None, gx_amax: Optional[Amax] = None, amax_channels: Optional[int] = None,
               mask_bits: Optional[torch.Tensor] = None, gx_chmax: Optional[torch.Tensor] = None) -> torch.Tensor:
    """gx (+)= conv_transpose(gy, weight); gy must already carry the activation derivative.
    mask/nmask: afterwards gx[:, :nmask] *= LeakyReLU'(mask[:, :nmask]) in the same launch (mask = the saved
    activation that produced this conv's input), i.e. gx comes out as a PRE-activation gradient.
    res/alpha: gx = res + alpha * conv_transpose(...) (residual branches: the skip gradient is added in the epilogue).
    gy_amax / gx_amax: as x_amax / y_amax of conv_forward (gx_amax bounds the COMPLETE gx: res, accumulate and mask included).
    amax_channels (Cout <= 2 heads only): gx_amax bounds gx[:, :amax_channels] instead of all of gx.
    mask_bits (with mask / nmask): the bits conv_forward(bits_out=...) wrote for `mask`; used instead of the fp32 tensor when this data
    gradient runs on the fp16x2 streaming kernel (both layers passed x3s_bits_ok), ignored otherwise.
    gx_chmax: a ZEROED (Cin,) tensor that holds max |gx[:, c]| per channel of the complete gx after the call: folded by the launch's
    epilogue on the fp16x2 kernels (irr_conv_x3_next_chmax), one pass over gx on every other route -- the scales of
    the weight gradient that takes gx as its gy (conv_wgrad(gy_chmax=...))."""
    B, cout, oh, ow = gy.shape
    cout_w, cin, k, _ = weight.shape
    assert cout == cout_w
    H, W = in_hw
    if gx is None:
        gx = torch.empty(B, cin, H, W, device=gy.device, dtype=torch.float32)
        accumulate = False
    if (res is not None or alpha != 1.0) and stride != 1:
        raise ValueError("res/alpha epilogue is only wired for stride-1 data gradients")
    margs = (hip.ptr(mask), hip.bs(mask), int(nmask)) if (mask is not None and nmask > 0) else (None, 0, 0)
    if stride == 1 and cout <= 2 and k == 3 and res is None and alpha == 1.0:
        # tiny-Cout heads: a pure HBM stream over the Cin-channel gradient buffer (VALU kernel, csrc/conv_small.hip)
        wc = weight.detach().contiguous()
        LAUNCHES["dgrad_smallco"] += 1
        # gx_amax here bounds gx[:, :amax_channels] (default: all of gx), folded by the same pass
        nam = cin if amax_channels is None else int(amax_channels)
        hip.call("irr_conv2d_smallco_dgrad_f32", hip.ptr(gy), hip.ptr(wc), hip.ptr(gx), margs[0], B, cin, H, W, cout, dil,
                 hip.bs(gy), hip.bs(gx), margs[1], margs[2], int(accumulate),
                 gx_amax.ptr() if gx_amax is not None else None, nam if gx_amax is not None else 0, hip.stream())
        if gx_chmax is not None:                            # (this kernel folds no channel maxima)
            channel_amax(gx, gx_chmax)
        return gx
    if stride == 1 and cout == 1:
        # the MFMA kernel consumes input channels in pairs: give the single-channel gradient a zero partner
        gy = torch.cat([gy, torch.zeros_like(gy)], dim=1)
        weight = torch.cat([weight.detach(), torch.zeros_like(weight.detach())], dim=0)
        cout = 2
    h2 = False
    use_bits = False
    if stride == 1 and cout >= 2:
        code = x3_code(B, cout, oh, ow, cin, k, 1, dil)
        h2 = bool(code) and bool(h2_code(B, cout, oh, ow, cin, k, 1, dil))
        if code:
            wq = packed_weights_h2(weight, True) if h2 else packed_weights_x3(weight, True)
            args = ("irr_conv2d_fwd_x3", hip.ptr(gy), hip.ptr(wq), None, hip.ptr(res), hip.ptr(gx), B, cout, oh, ow, cin,
                    dil, hip.bs(gy), hip.bs(gx), hip.bs(res) if res is not None else 0, 0, float(alpha),
                    int(accumulate), *margs, hip.stream())
            if h2:
                args, _ga = _h2_args(args, gy, gy_amax, None if (code == 9001 and _X3S_NO_FUSED_AMAX) else gx_amax)
            use_bits = mask_bits is not None and margs[2] > 0 and h2 and code == 9001 and cin <= 32 and not _X3S_BITS_NOREAD
            if use_bits:
                args = ("irr_conv2d_fwd_h2_bits",) + tuple(args[1:-7]) + (mask_bits.data_ptr(), margs[2], None) + tuple(args[-4:])
            variant = (x3s_variant(h2, res, accumulate, mask is not None and nmask > 0, bits=use_bits) if code == 9001
                       else (200000 if h2 else 100000) + code)
            LAUNCHES["dgrad_x3s" if code == 9001 else "dgrad_h2" if h2 else "dgrad_x3"] += 1
        else:
            LAUNCHES["dgrad_f32"] += 1
            wp = packed_weights(weight, True)
            args = ("irr_conv2d_fwd_f32", hip.ptr(gy), hip.ptr(wp), None, hip.ptr(res), hip.ptr(gx), B, cout, oh, ow, cin, H, W,
                    k, 1, dil, hip.bs(gy), hip.bs(gx), hip.bs(res) if res is not None else 0, 0, float(alpha),
                    int(accumulate), *margs, hip.stream())
            variant = None
        fold_ch = gx_chmax is not None and h2 and not (code == 9001 and (_X3S_NO_FUSED_AMAX or _X3S_NO_CH_FOLD))     # (either kernel family folds them in its epilogue)
        if fold_ch:
            hip.lib().irr_conv_x3_next_chmax(gx_chmax.data_ptr())      # (one-shot: the launch below)
            gx_chmax = None
        if TIMER is None:
            _call_conv(args)
        else:
            if variant is None:
                variant = hip.lib().irr_conv2d_fwd_variant(B, cin, H, W, k)
            TIMER.wrap(variant, 2.0 * B * H * W * cout * (real_cin or cin) * k * k, lambda: _call_conv(args), "dgrad",
                       nbytes=_map_bytes(B, oh, ow, cout) + _map_bytes(B, H, W, cin * (1 + (res is not None) + bool(accumulate))
                                                                       + ((1 if use_bits else min(nmask, cin)) if mask is not None else 0)))
    elif stride == 2 and k == 3 and dil == 1 and H == 2 * oh and W == 2 * ow and cout >= 2 and cin > S2_GATHER_MAX_CIN:
        # transposed stride-2 conv == stride-1 conv (flipped weights) of the zero-interleaved gradient
        z = torch.zeros(B, cout, H, W, device=gy.device, dtype=torch.float32)
        z[:, :, ::2, ::2] = gy
        wp = packed_weights(weight, True)
        hip.call("irr_conv2d_fwd_f32", hip.ptr(z), hip.ptr(wp), None, None, hip.ptr(gx), B, cout, H, W, cin, H, W,
                 k, 1, 1, hip.bs(z), hip.bs(gx), 0, 0, 1.0, int(accumulate), *margs, hip.stream())
    else:
        # stride-2 3x3 layers (the feature pyramid, incl. the image gradient of its first conv): a 2x2-block kernel over the four
        # parity classes of the transposed conv; the MFMA route above over a zero-interleaved copy of gy spends 75 % of its
        # work on zeros (tools/s2_dgrad_bench.py)
        LAUNCHES["dgrad_strided"] += 1
        tmp = gx if not accumulate else torch.empty(B, cin, H, W, device=gy.device, dtype=torch.float32)
        wc = weight.detach().contiguous()
        hip.call("irr_conv2d_dgrad_strided_f32", hip.ptr(gy), hip.ptr(wc), hip.ptr(tmp), B, cin, H, W, cout, oh, ow,
                 k, stride, dil, hip.bs(gy), hip.bs(tmp), hip.stream())
        if accumulate:
            gx += tmp
        if mask is not None and nmask > 0:
            gx[:, :nmask] *= torch.where(mask[:, :nmask] > 0, 1.0, 0.1)
    if gx_amax is not None and (not h2 or (code == 9001 and _X3S_NO_FUSED_AMAX)):
        amax_measure(gx, gx_amax)
    if gx_chmax is not None:                                # (no launch folded them)
        channel_amax(gx, gx_chmax)
    if _CHECK_FINITE and _CHECK_FINITE != "slots":
        _check_finite(f"conv_dgrad {tuple(gy.shape)} -> {cin} dil {dil} stride {stride} h2 {h2}", gx, gy_amax, gx_amax)
    return gx



# one operand scale per channel for the weight gradient's gy-role operand (round 6; IRR_WGRAD_CHANNEL_SCALE=0: the tensor's scale, A/B)
WGRAD_CHANNEL_SCALE = os.environ.get("IRR_WGRAD_CHANNEL_SCALE", "1") != "0"


# tensors above this size keep the tensor's one scale when nobody folded their channel maxima on the way (the 32-channel full-resolution
# maps of the occlusion upsampler: 1.4 GB each, seven per call -- a pass over them would cost 2 ms per step)
WGRAD_CHANNEL_PASS_MAX_BYTES = int(float(os.environ.get("IRR_WGRAD_CH_PASS_MAX_MB", "768")) * (1 << 20))


CHANNEL_PASS_LOG = None


def channel_amax(t: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """max |t[:, c]| per channel of a plane-dense (B, C, H, W) tensor: (C,) float32, one HBM pass (irr_amax_channels_f32)"""
    B, C, H, W = t.shape
    if out is None:
        out = zero_slots(t.device, C)                       # (pre-zeroed pool memory: no fill launch per call)
    assert out.numel() == C and out.is_contiguous()
    LAUNCHES["amax_channels"] += 1
    if CHANNEL_PASS_LOG is not None:                        # (tools/r6_chs_census.py: which tensors still need a pass)
        import traceback
        CHANNEL_PASS_LOG.append((tuple(t.shape), [f"{f.name}:{f.lineno}" for f in traceback.extract_stack(limit=8)][:-1]))
    hip.call("irr_amax_channels_f32", hip.ptr(t), B, C, H * W, hip.bs(t), hip.ptr(out), 1, hip.stream())
    return out


def zero_slots(device, n: int) -> torch.Tensor:
    """n zeroed float32 values for a launch to fold maxima into -- carved from the amax pool (one fill per pool, not per vector)"""
    a = Amax.zeros(device, n)
    return a.slots[a.first:a.first + a.n]


def conv_wgrad(x: torch.Tensor, gy: torch.Tensor, weight_shape, stride: int, dil: int,
               gw: Optional[torch.Tensor] = None, gbias: Optional[torch.Tensor] = None, alpha: float = 1.0,
               defer: Optional[ReduceBatch] = None, x_amax: Optional[Amax] = None, gy_amax: Optional[Amax] = None,
               x_chmax: Optional[torch.Tensor] = None, gy_chmax: Optional[torch.Tensor] = None) -> torch.Tensor:
    """gw += d/dW; gw (Cout,Cin,k,k) is created zeroed when not given.  gbias (optional, (Cout,)) += sum of gy over
    (b, h, w): the bias gradient comes out of the same launch (the gy tiles are staged there anyway).
    ``defer``: the MFMA kernels leave the fold of their partial images to ``defer.run()`` (gw is complete only after it).
    x_chmax / gy_chmax (fp16x2 route): max |.| per channel of x / gy when a producer has folded them (conv_dgrad(gx_chmax=...)); the
    operand in the kernel's gy role is scaled channel by channel from them -- measured here by one pass when absent and the operand
    is not larger than WGRAD_CHANNEL_PASS_MAX_BYTES (above that: the tensor's one scale)."""
    cout, cin, k, _ = weight_shape
    B, _, H, W = x.shape
    _, _, oh, ow = gy.shape
    if gw is None:
        gw = torch.zeros(cout, cin, k, k, device=x.device, dtype=torch.float32)
    assert gw.is_contiguous()
    use_x3 = (MATH in ("x3", "h2") and not (cout <= 4 and stride == 1)
              and bool(hip.lib().irr_conv2d_wgrad_x3_eligible(B, cin, H, W, cout, k, stride, dil)))
    # h2: where both operands carry amax slots (the callers provide them for the layers whose forward runs on the h2 kernel)
    use_h2 = use_x3 and MATH == "h2" and x_amax is not None and gy_amax is not None
    # scratch: one partial [Cout][k*k][Cin] image per block column of the launch (the Cout <= 4 / Cin = 3 kernels: one image)
    smallci = cin == 3 and k == 3 and cout > 4
    if (cout <= 4 and stride == 1) or smallci:
        nws = cout * cin * k * k
    elif use_x3:
        nws = hip.lib().irr_conv2d_wgrad_x3_ws_elems(cin, cout)
    else:
        nws = hip.lib().irr_conv2d_wgrad_ws_elems(B, cin, H, W, cout, oh, ow, k, stride, dil, hip.bs(x), hip.bs(gy))
    ws = torch.empty(nws, device=x.device, dtype=torch.float32)
    LAUNCHES["wgrad_smallci" if smallci else "wgrad_smallco" if (cout <= 4 and stride == 1) else "wgrad_h2" if use_h2 else
             "wgrad_x3_dil" if (use_x3 and dil > 1) else "wgrad_x3" if use_x3 else "wgrad_f32"] += 1
    if smallci:
        hip.call("irr_conv2d_smallci_wgrad_f32", hip.ptr(x), hip.ptr(gy), hip.ptr(gw), hip.ptr(ws), hip.ptr(gbias), float(alpha), B, cin, H, W,
                 cout, oh, ow, stride, dil, hip.bs(x), hip.bs(gy), hip.stream())
        return gw
    if cout <= 4 and stride == 1:
        hip.call("irr_conv2d_smallco_wgrad_f32", hip.ptr(x), hip.ptr(gy), hip.ptr(gw), hip.ptr(ws), hip.ptr(gbias), float(alpha), B, cin, H, W, cout, k, dil,
                 hip.bs(x), hip.bs(gy), hip.stream())
        return gw
    if defer is not None:
        defer.begin()
    try:
        if use_h2 and WGRAD_CHANNEL_SCALE:
            # one scale per channel for the operand in the kernel's gy role (the other one is robust element by element): its channel
            # maxima from one pass over it, unless the caller has them
            robust_x = bool(hip.lib().irr_conv2d_wgrad_h2_robust_side(B, cin, H, W, cout, dil))
            role = gy if robust_x else x
            chm = gy_chmax if robust_x else x_chmax
            if callable(chm):                                # (measured lazily, on the stream this launch runs on: the DenseNet node's buffer maxima)
                chm = chm()
            if chm is None and role.numel() * 4 <= WGRAD_CHANNEL_PASS_MAX_BYTES:
                chm = channel_amax(role)
        if use_h2 and WGRAD_CHANNEL_SCALE and chm is not None:
            assert chm.numel() == role.shape[1] and chm.is_contiguous() and chm.dtype == torch.float32
            hip.call("irr_conv2d_wgrad_h2_ch", hip.ptr(x), hip.ptr(gy), hip.ptr(gw), hip.ptr(ws), hip.ptr(gbias), float(alpha), B, cin, H, W,
                     cout, dil, hip.bs(x), hip.bs(gy), x_amax.ptr(), x_amax.n, gy_amax.ptr(), gy_amax.n,
                     None if robust_x else chm.data_ptr(), chm.data_ptr() if robust_x else None, hip.stream())
        elif use_h2:
            hip.call("irr_conv2d_wgrad_h2", hip.ptr(x), hip.ptr(gy), hip.ptr(gw), hip.ptr(ws), hip.ptr(gbias), float(alpha), B, cin, H, W,
                     cout, dil, hip.bs(x), hip.bs(gy), x_amax.ptr(), x_amax.n, gy_amax.ptr(), gy_amax.n, hip.stream())
        elif use_x3 and dil > 1:
            hip.call("irr_conv2d_wgrad_x3_dil", hip.ptr(x), hip.ptr(gy), hip.ptr(gw), hip.ptr(ws), hip.ptr(gbias), float(alpha), B, cin, H, W,
                     cout, dil, hip.bs(x), hip.bs(gy), hip.stream())
        elif use_x3:
            hip.call("irr_conv2d_wgrad_x3", hip.ptr(x), hip.ptr(gy), hip.ptr(gw), hip.ptr(ws), hip.ptr(gbias), float(alpha), B, cin, H, W,
                     cout, hip.bs(x), hip.bs(gy), hip.stream())
        else:
            hip.call("irr_conv2d_wgrad_f32", hip.ptr(x), hip.ptr(gy), hip.ptr(gw), hip.ptr(ws), hip.ptr(gbias), float(alpha), B, cin, H, W, cout, oh, ow, k, stride, dil,
                     hip.bs(x), hip.bs(gy), nws, hip.stream())
    finally:
        if defer is not None:
            defer.end(ws, gw)
    return gw



# the installed weight-gradient lane (None: weight gradients go through autograd)
SIDE: Optional[WgradSide] = None

# the autograd nodes (import at the end: they use the primitives above and read SIDE / TIMER through this module)
from .conv_nodes import (CAT_MAX_PARTS, _ConvBlock, _ConvChainFn, _DenseEstimatorFn, _OccUpsampleFn, _planes_dense, cat_channels, cat_channels_into,  # noqa: E402,F401
                         conv_block, conv_chain, dense_estimator, lrelu_bwd_bias, occ_upsample_net, wgrad_param)
