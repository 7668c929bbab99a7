"""conv() blocks of IRR-PWC on the MFMA kernels of libirr_hip.so: the fp32-faithful bf16x3-split family (conv_x3 /
conv_x3s / conv_wgrad_x3) wherever its ``*_eligible`` predicates accept the problem, the fp32-MFMA family
elsewhere (``IRR_CONV_MATH=f32`` forces the latter everywhere).

Mirrors the reference helper ``conv(in_planes, out_planes, kernel_size, stride, dilation, isReLU)``
(models/pwc_modules.py:8-19, models/irr_modules.py:7-18): Conv2d with "same" padding and bias,
optionally followed by LeakyReLU(0.1).  All tensors may be channel-slice views of larger NCHW buffers
(dense H*W planes, arbitrary batch stride), which is how the DenseNet decoders avoid ``torch.cat``.

There is no other backend in this module: every launch goes to libirr_hip.so (an A/B harness against torch's GPU
convolution lives in tools/torch_conv_backend.py, outside the product).
"""
from __future__ import annotations

import collections
import ctypes
import os
import weakref
from typing import Optional, Tuple

import torch

from . import hip

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

    def wrap(self, variant: int, flops: float, launch, phase: str = "fwd"):
        if not self.active:
            launch()
            return
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        launch()
        e.record()
        self.records.append((variant, flops, s, e, phase))

    def summary(self):
        """{variant: {calls, flops, seconds, fwd_calls, fwd_flops, fwd_seconds}} -- the fwd_* entries cover only the
        forward-pass launches, which never share the chip with the asynchronous weight-gradient lane."""
        torch.cuda.synchronize()
        agg = {}
        for variant, flops, s, e, phase in self.records:
            a = agg.setdefault(variant, [0, 0.0, 0.0, 0, 0.0, 0.0])
            dt = s.elapsed_time(e) * 1e-3
            a[0] += 1
            a[1] += flops
            a[2] += dt
            if phase == "fwd":
                a[3] += 1
                a[4] += flops
                a[5] += dt
        self.records.clear()
        return {v: {"calls": a[0], "flops": a[1], "seconds": a[2], "fwd_calls": a[3], "fwd_flops": a[4],
                    "fwd_seconds": a[5]} for v, a in agg.items()}


TIMER: Optional[KernelTimer] = None

# launches per kernel family since the last clear() -- what a step was ROUTED to (tests assert on it, bench.py reports it)
LAUNCHES: "collections.Counter[str]" = collections.Counter()


def out_hw(h: int, w: int, k: int, stride: int, dil: int) -> Tuple[int, int]:
    pad = ((k - 1) * dil) // 2
    return ((h + 2 * pad - dil * (k - 1) - 1) // stride + 1, (w + 2 * pad - dil * (k - 1) - 1) // stride + 1)


# ----------------------------------------------------------------------------------------------
# packed-weight cache: weights are re-packed only when the parameter changed (optimizer step), and then ALL AT ONCE
# ----------------------------------------------------------------------------------------------
# bumped by anything that rewrites parameters outside autograd's version counters (FusedAdam.step, TrainStep)
WEIGHT_EPOCH = [0]


class _PackRegistry:
    """Every packed copy of a live conv weight on one device, as a job of the batched pack launch (csrc/pack_batch.hip).

    A copy is registered the first time it is built (single-job launch).  When the weight epoch changes (optimizer step) the
    first cache miss refreshes EVERY registered copy with one dispatch and re-tags the caches, so a train step issues one pack
    launch instead of ~250.  Entries hold weak references: they disappear with their model."""

    def __init__(self, device):
        self.device = device
        self.entries = {}               # key -> (weakref(weight), dst tensor, builder(job_addr, w_ptr) -> nblocks, retag())
        self.version = 0
        self.epoch = WEIGHT_EPOCH[0]
        self._table = None              # (signature, device table, njobs, nblocks)

    def register(self, key, weight, dst, builder, retag):
        if key in self.entries:
            return
        reg = self

        def _gone(_ref, key=key):
            if reg.entries.pop(key, None) is not None:
                reg.version += 1
        self.entries[key] = (weakref.ref(weight, _gone), dst, builder, retag)
        self.version += 1

    def refresh(self) -> bool:
        """called on a cache miss: if the epoch moved since the last refresh, repack everything registered (True)"""
        if self.epoch == WEIGHT_EPOCH[0] or not self.entries:
            self.epoch = WEIGHT_EPOCH[0]
            return False
        self.epoch = WEIGHT_EPOCH[0]
        live = [(k, e, e[0]()) for k, e in list(self.entries.items())]
        live = [(k, e, w) for k, e, w in live if w is not None and w.is_contiguous()]
        if not live:
            return False
        sig = (self.version, tuple(w.data_ptr() for _, _, w in live))
        if self._table is None or self._table[0] != sig:
            jb = hip.lib().irr_conv_pack_job_bytes()
            b0 = hip.lib().irr_conv_pack_job_block0_offset()
            buf = ctypes.create_string_buffer(jb * len(live))
            base = ctypes.addressof(buf)
            block0 = 0
            for n_, (_, e, w) in enumerate(live):
                nb = e[2](base + n_ * jb, w.data_ptr())
                if nb < 0:
                    raise hip.HipError(f"pack job rejected ({nb})")
                ctypes.c_long.from_address(base + n_ * jb + b0).value = block0
                block0 += nb
            host = torch.frombuffer(buf, dtype=torch.uint8).clone()
            self._table = (sig, host.to(self.device), len(live), block0)
        _, table, njobs, nblocks = self._table
        with hip.device_of(table):
            hip.call("irr_conv_pack_batch", hip.ptr(table), njobs, nblocks, hip.stream())
        LAUNCHES["pack_batch"] += 1
        for _, e, _ in live:
            e[3]()
        return True


_REGISTRIES = {}


def _registry(device) -> _PackRegistry:
    r = _REGISTRIES.get(device.index)
    if r is None:
        r = _REGISTRIES[device.index] = _PackRegistry(device)
    return r


def _weight_tag(w: torch.Tensor):
    return (w.data_ptr(), w._version, tuple(w.shape), WEIGHT_EPOCH[0])


def _announce_rewrite(reg: _PackRegistry, old_tag, new_tag) -> None:
    """A cached packed copy is stale although nobody moved the weight epoch: the parameter was rewritten in place by code that
    does not know about the caches -- ``torch.optim.Adam.step()`` under the reference's own training loop (runtime.py:189),
    ``load_state_dict``.  Treat it as an optimizer step: move the epoch so that the registry refreshes EVERY packed copy with its
    one batched launch instead of ~250 single-job launches trickling in layer by layer."""
    if old_tag[3] == new_tag[3] and reg.epoch == WEIGHT_EPOCH[0] and old_tag[:3] != new_tag[:3]:
        WEIGHT_EPOCH[0] += 1


def _packed(weight: torch.Tensor, transpose: bool, slot: str, nbytes_fn, dtype, single, builder_name):
    """shared body of packed_weights / packed_weights_x3: cache ON the tensor object (so it dies with the parameter and can
    never be confused with another tensor that later reuses the same address), refreshed whenever the parameter's storage,
    version counter or the weight epoch changes -- through the batched launch when the copy is already registered."""
    cache = weight.__dict__.setdefault(slot, {})
    w = weight.detach()
    key = bool(transpose)
    tag = _weight_tag(w)
    hit = cache.get(key)
    if hit is not None and hit[0] == tag:
        return hit[1]
    reg = _registry(w.device)
    if hit is not None:
        if not weight.__dict__.get("_irr_derived", False):
            _announce_rewrite(reg, hit[0], tag)
        if reg.refresh():
            hit = cache.get(key)
            if hit[0] == _weight_tag(w):
                return hit[1]
        tag = _weight_tag(w)
    cout, cin, k, _ = w.shape
    lcin, lcout = (cout, cin) if transpose else (cin, cout)
    n = nbytes_fn(lcin, lcout, k)
    wp = hit[1] if (hit is not None and hit[1].numel() == n and hit[1].device == w.device) else \
        torch.empty(n, device=w.device, dtype=dtype)
    wc = w.contiguous()
    single(wc, wp, lcin, lcout, k, int(transpose))
    LAUNCHES["pack_single"] += 1
    cache[key] = (tag, wp)
    if w.is_contiguous():
        wref = weakref.ref(weight)

        def retag(cache=cache, key=key, wp=wp, wref=wref):
            t = wref()
            if t is not None:
                cache[key] = (_weight_tag(t.detach()), wp)
        fn = getattr(hip.lib(), builder_name)
        if builder_name == "irr_conv_pack_job_f32":
            builder = lambda job, wptr, wp=wp: fn(job, wptr, wp.data_ptr(), lcin, lcout, k, int(transpose))
        else:
            builder = lambda job, wptr, wp=wp: fn(job, wptr, wp.data_ptr(), lcin, lcout, int(transpose))
        reg.register((id(weight), slot, key), weight, wp, builder, retag)
    return wp


def packed_weights(weight: torch.Tensor, transpose: bool) -> torch.Tensor:
    """Packed copy of ``weight`` for irr_conv2d_fwd_f32 (see _packed)."""
    return _packed(weight, transpose, "_irr_packed", lambda ci, co, k: hip.lib().irr_conv_packed_weight_elems(ci, co, k),
                   torch.float32,
                   lambda wc, wp, ci, co, k, tr: hip.call("irr_conv_pack_weights_f32", hip.ptr(wc), hip.ptr(wp), ci, co, k, tr,
                                                          hip.stream()),
                   "irr_conv_pack_job_f32")


# "x3": 3x3 stride-1 convs run on the bf16 matrix pipe with exact 3-way operand splits (csrc/conv_x3.hip, fp32-faithful)
# wherever irr_conv2d_x3_eligible accepts the problem; "f32": the fp32-MFMA kernel everywhere (A/B runs).
MATH = os.environ.get("IRR_CONV_MATH", "x3")


def set_math(name: str) -> None:
    global MATH
    if name not in ("x3", "f32"):
        raise ValueError(name)
    MATH = name


_X3_ENV_DONE = [False]


def x3_code(B: int, cin: int, H: int, W: int, cout: int, k: int, stride: int, dil: int) -> int:
    if MATH != "x3":
        return 0
    if not _X3_ENV_DONE[0]:
        _X3_ENV_DONE[0] = True
        if os.environ.get("IRR_X3_MIN_BLOCKS"):          # tests: exercise the kernel on small problems too
            hip.lib().irr_conv_x3_set_min_blocks(int(os.environ["IRR_X3_MIN_BLOCKS"]))
    return int(hip.lib().irr_conv2d_x3_eligible(B, cin, H, W, cout, k, stride, dil))


def packed_weights_x3(weight: torch.Tensor, transpose: bool) -> torch.Tensor:
    """Pre-split (3 x bf16) packed copy of ``weight`` for irr_conv2d_fwd_x3 (see _packed)."""
    assert weight.shape[2] == 3
    return _packed(weight, transpose, "_irr_packed_x3", lambda ci, co, k: hip.lib().irr_conv_x3_packed_bytes(ci, co), torch.uint8,
                   lambda wc, wp, ci, co, k, tr: hip.call("irr_conv_pack_weights_x3", hip.ptr(wc), hip.ptr(wp), ci, co, tr,
                                                          hip.stream()),
                   "irr_conv_pack_job_x3")


# ----------------------------------------------------------------------------------------------
# primitives (no autograd)
# ----------------------------------------------------------------------------------------------
def _call_conv(args) -> None:
    """Launch a conv entry point from its argument tuple.  irr_conv2d_fwd_x3 problems that are too small to fill the chip
    get a scratch buffer and run through the K-split entry point (csrc/conv_x3.hip: blockIdx.z splits the channel chunks)."""
    if args[0] == "irr_conv2d_fwd_x3":
        B, cin, H, W, cout, dil = args[6:12]
        n = hip.lib().irr_conv2d_fwd_x3_ws_elems(B, cin, H, W, cout, dil)
        if n > 0:
            ws = torch.empty(n, dtype=torch.float32, device=torch.device("cuda", torch.cuda.current_device()))
            hip.call("irr_conv2d_fwd_x3_splitk", *args[1:-1], ws.data_ptr(), n, args[-1])
            return
    hip.call(*args)


def conv_forward(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], stride: int, dil: int,
                 lrelu: bool, out: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None,
                 alpha: float = 1.0, accumulate: bool = False, real_cin: Optional[int] = None) -> torch.Tensor:
    """out = [res +] alpha * act(conv(x, weight) + bias)   (accumulate: out += ...).
    real_cin: the layer's true input-channel count when x / weight are zero-padded copies (KernelTimer prices algorithmic FLOPs)."""
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
        return out
    code = x3_code(B, cin, H, W, cout, k, stride, dil)
    if code:
        wq = packed_weights_x3(weight, False)
        args = ("irr_conv2d_fwd_x3", hip.ptr(x), hip.ptr(wq), hip.ptr(bias.detach() if bias is not None else None),
                hip.ptr(res), hip.ptr(out), B, cin, H, W, cout, dil,
                hip.bs(x), hip.bs(out), hip.bs(res) if res is not None else 0,
                int(lrelu), float(alpha), int(accumulate), None, 0, 0, hip.stream())
        variant = 100000 + code
        LAUNCHES["fwd_x3s" if code == 9001 else "fwd_x3"] += 1
    else:
        LAUNCHES["fwd_f32"] += 1
        wp = packed_weights(weight, False)
        args = ("irr_conv2d_fwd_f32", hip.ptr(x), hip.ptr(wp), hip.ptr(bias.detach() if bias is not None else None),
                hip.ptr(res), hip.ptr(out), B, cin, H, W, cout, oh, ow, k, stride, dil,
                hip.bs(x), hip.bs(out), hip.bs(res) if res is not None else 0,
                int(lrelu), float(alpha), int(accumulate), None, 0, 0, hip.stream())
        variant = None
    if TIMER is None:
        _call_conv(args)
    else:
        if variant is None:
            variant = hip.lib().irr_conv2d_fwd_variant(B, cout, oh, ow, k)
        TIMER.wrap(variant, 2.0 * B * oh * ow * cout * (real_cin or cin) * k * k, lambda: _call_conv(args))
    return out


def conv_forward_skip(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], lrelu: bool, skip: torch.Tensor):
    """(e, y) with e = act(conv3x3(x) + bias) and y = skip + e.  On the streaming 32-channel kernel both come out of ONE launch
    (irr_conv2d_fwd_x3_dual); elsewhere e is computed and the sum is an elementwise pass."""
    B, cin, H, W = x.shape
    cout = weight.shape[0]
    if x3_code(B, cin, H, W, cout, 3, 1, 1) == 9001 and _planes_dense(skip) and not os.environ.get("IRR_X3S_NO_DUAL"):      # (A/B switch)
        e = torch.empty(B, cout, H, W, device=x.device, dtype=torch.float32)
        y = torch.empty_like(e)
        wq = packed_weights_x3(weight, False)
        LAUNCHES["fwd_x3s"] += 1
        args = ("irr_conv2d_fwd_x3_dual", hip.ptr(x), hip.ptr(wq), hip.ptr(bias.detach() if bias is not None else None), hip.ptr(skip),
                hip.ptr(y), hip.ptr(e), B, cin, H, W, cout, 1, hip.bs(x), hip.bs(y), hip.bs(skip), hip.bs(e), int(lrelu), 1.0, hip.stream())
        if TIMER is None:
            hip.call(*args)
        else:
            TIMER.wrap(100000 + 9001, 2.0 * B * H * W * cout * cin * 9, lambda: hip.call(*args))
        return e, y
    e = conv_forward(x, weight, bias, 1, 1, lrelu)
    return e, torch.add(skip, e)


S2_GATHER_MAX_CIN = 96   # stride-2 3x3 data gradients with at most this many result channels use the 2x2-block kernel
                         # (measured at the BASELINE batch: faster up to 128 -> 96 at 24x28, slower for 196 -> 128 at 12x14; A/B: 0)


def conv_dgrad(gy: torch.Tensor, weight: torch.Tensor, stride: int, dil: int, in_hw: Tuple[int, int],
               gx: Optional[torch.Tensor] = None, accumulate: bool = False,
               mask: Optional[torch.Tensor] = None, nmask: int = 0,
               res: Optional[torch.Tensor] = None, alpha: float = 1.0, real_cin: Optional[int] = None) -> torch.Tensor:
    """gx (+)= conv_transpose(gy, weight); gy must already carry the activation derivative.
    mask/nmask: afterwards gx[:, :nmask] *= LeakyReLU'(mask[:, :nmask]) in the same launch (mask = the saved
    activation that produced this conv's input), i.e. gx comes out as a PRE-activation gradient.
    res/alpha: gx = res + alpha * conv_transpose(...) (residual branches: the skip gradient is added in the epilogue)."""
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
        hip.call("irr_conv2d_smallco_dgrad_f32", hip.ptr(gy), hip.ptr(wc), hip.ptr(gx), margs[0], B, cin, H, W, cout, dil,
                 hip.bs(gy), hip.bs(gx), margs[1], margs[2], int(accumulate), hip.stream())
        return gx
    if stride == 1 and cout == 1:
        # the MFMA kernel consumes input channels in pairs: give the single-channel gradient a zero partner
        gy = torch.cat([gy, torch.zeros_like(gy)], dim=1)
        weight = torch.cat([weight.detach(), torch.zeros_like(weight.detach())], dim=0)
        cout = 2
    if stride == 1 and cout >= 2:
        code = x3_code(B, cout, oh, ow, cin, k, 1, dil)
        if code:
            wq = packed_weights_x3(weight, True)
            args = ("irr_conv2d_fwd_x3", hip.ptr(gy), hip.ptr(wq), None, hip.ptr(res), hip.ptr(gx), B, cout, oh, ow, cin,
                    dil, hip.bs(gy), hip.bs(gx), hip.bs(res) if res is not None else 0, 0, float(alpha),
                    int(accumulate), *margs, hip.stream())
            variant = 100000 + code
            LAUNCHES["dgrad_x3s" if code == 9001 else "dgrad_x3"] += 1
        else:
            LAUNCHES["dgrad_f32"] += 1
            wp = packed_weights(weight, True)
            args = ("irr_conv2d_fwd_f32", hip.ptr(gy), hip.ptr(wp), None, hip.ptr(res), hip.ptr(gx), B, cout, oh, ow, cin, H, W,
                    k, 1, dil, hip.bs(gy), hip.bs(gx), hip.bs(res) if res is not None else 0, 0, float(alpha),
                    int(accumulate), *margs, hip.stream())
            variant = None
        if TIMER is None:
            _call_conv(args)
        else:
            if variant is None:
                variant = hip.lib().irr_conv2d_fwd_variant(B, cin, H, W, k)
            TIMER.wrap(variant, 2.0 * B * H * W * cout * (real_cin or cin) * k * k, lambda: _call_conv(args), "dgrad")
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
    return gx


class ReduceBatch:
    """Fold jobs of weight-gradient launches whose partial images have not been added to their gradients yet
    (include/irr_hip.h, "deferred fold"): the MFMA weight-gradient launchers append a job here instead of running their own
    10-20 us fold kernel, and ``run()`` folds all of them with ONE launch.  Holds the scratch tensors alive until then."""

    def __init__(self):
        lib = hip.lib()
        self.jb = lib.irr_wgrad_job_bytes()
        self.cap = lib.irr_wgrad_reduce_batch_max()
        self.buf = ctypes.create_string_buffer(self.jb * self.cap)
        self.n = 0
        self.keep = []                   # scratch (and gradient) tensors of the pending jobs
        self.targets = set()             # data_ptr of the gradients with a pending job: a batch folds into each at most once

    def begin(self):
        hip.lib().irr_wgrad_defer_begin(ctypes.addressof(self.buf) + self.n * self.jb, self.cap - self.n)

    def end(self, ws: torch.Tensor, gw: torch.Tensor):
        got = hip.lib().irr_wgrad_defer_end()
        if got:
            self.n += got
            self.keep += [ws, gw]
            self.targets.add(gw.data_ptr())

    def full_for(self, gw: torch.Tensor) -> bool:
        return self.n >= self.cap - 1 or gw.data_ptr() in self.targets

    def run(self):
        """launch the fold of every pending job on the current stream; returns the tensors that must outlive it"""
        keep = self.keep
        if self.n:
            with hip.device_of(keep[0]):
                hip.call("irr_wgrad_reduce_batch", ctypes.addressof(self.buf), self.n, hip.stream())
            LAUNCHES["wgrad_reduce_batch"] += 1
        self.n, self.keep, self.targets = 0, [], set()
        return keep


def conv_wgrad(x: torch.Tensor, gy: torch.Tensor, weight_shape, stride: int, dil: int,
               gw: Optional[torch.Tensor] = None, gbias: Optional[torch.Tensor] = None, alpha: float = 1.0,
               defer: Optional[ReduceBatch] = None) -> torch.Tensor:
    """gw += d/dW; gw (Cout,Cin,k,k) is created zeroed when not given.  gbias (optional, (Cout,)) += sum of gy over
    (b, h, w): the bias gradient comes out of the same launch (the gy tiles are staged there anyway).
    ``defer``: the MFMA kernels leave the fold of their partial images to ``defer.run()`` (gw is complete only after it)."""
    cout, cin, k, _ = weight_shape
    B, _, H, W = x.shape
    _, _, oh, ow = gy.shape
    if gw is None:
        gw = torch.zeros(cout, cin, k, k, device=x.device, dtype=torch.float32)
    assert gw.is_contiguous()
    use_x3 = (MATH == "x3" and not (cout <= 4 and stride == 1)
              and bool(hip.lib().irr_conv2d_wgrad_x3_eligible(B, cin, H, W, cout, k, stride, dil)))
    # scratch: one partial [Cout][k*k][Cin] image per block column of the launch (the Cout <= 4 / Cin = 3 kernels: one image)
    smallci = cin == 3 and k == 3 and cout > 4
    if (cout <= 4 and stride == 1) or smallci:
        nws = cout * cin * k * k
    elif use_x3:
        nws = hip.lib().irr_conv2d_wgrad_x3_ws_elems(cin, cout)
    else:
        nws = hip.lib().irr_conv2d_wgrad_ws_elems(B, cin, H, W, cout, oh, ow, k, stride, dil, hip.bs(x), hip.bs(gy))
    ws = torch.empty(nws, device=x.device, dtype=torch.float32)
    LAUNCHES["wgrad_smallci" if smallci else "wgrad_smallco" if (cout <= 4 and stride == 1) else
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
        if use_x3 and dil > 1:
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


class WgradSide:
    """Asynchronous weight-gradient lane (training harness opt-in, see irr_amd.ddp.GradArena.enable_async_wgrad).

    dgrad and wgrad of a layer are independent once the pre-activation gradient exists, and only dgrad is on the
    critical path of backward.  With this object installed, every weight/bias gradient is accumulated straight into
    the flat gradient arena on a SECOND HIP stream: the wgrad kernels fill the SIMDs that the tail of a dgrad launch
    (or a whole coarse-level launch, which cannot fill 256 CUs) leaves idle.  Autograd then receives ``None`` for
    those parameters; GradArena.sync() joins the lane before the all-reduce / optimizer step."""

    def __init__(self, params_and_views, inline: bool = False):
        # id(parameter) -> (weak reference to the parameter, flat-arena view with its shape).  Looked up by id for speed and
        # verified by identity: the id of a dead parameter can be reused by a parameter of ANOTHER model
        self.views = {id(p_): (weakref.ref(p_), v_) for p_, v_ in params_and_views}
        dev = next(iter(self.views.values()))[1].device
        # inline: no second stream -- the launches stay on the current stream, but still accumulate straight into the arena
        # (no per-use gradient tensors, no autograd accumulation adds, one batched fold): GradArena.enable_direct_wgrad()
        self.inline = inline
        self.stream = None if inline else torch.cuda.Stream(device=dev)
        # The references in _inflight are dropped only after the lane has passed the launch (marker) or after the current stream
        # has joined the lane, so the caching allocator can never hand the memory out early; Tensor.record_stream on top of that
        # makes the allocator record one (system-scope) event on the lane per freed tensor -- 500 per step (A/B switch: 1 = on)
        self.record_streams = os.environ.get("IRR_LANE_RECORD_STREAM", "0") != "0"
        self._inflight = collections.deque()    # (done marker on the lane, tensors its launch reads)
        self.on_launch = None                   # optional hook(weight, bias) once a routed gradient is complete (ddp: early buckets)
        self.on_queue = None                    # optional hook(weight, bias) when a launch is queued (ddp: flush at a bucket's last one)
        # the ~210 partial-image folds of a step run as a few batched launches (ReduceBatch); IRR_LANE_BATCH_REDUCE=0: A/B
        self.batch = ReduceBatch() if os.environ.get("IRR_LANE_BATCH_REDUCE", "1") != "0" else None
        self._pending = []                      # (weight, bias) of launches whose fold has not been launched yet
        # Launches are handed to the lane in GROUPS: one event on the main stream + one wait on the lane per group instead of per
        # launch (~500 per step; every record / wait is a barrier packet in its queue).  Waiting for a LATER point of the main
        # stream than necessary is always safe -- nothing on the main stream writes what a queued launch reads (the tensors are
        # held alive here and the backward nodes never touch a gradient slice again once its weight-gradient launch is issued).
        self.group = max(1, int(os.environ.get("IRR_LANE_GROUP", "4")))
        self._queued = []                       # (fn, tensors, params) not handed to the lane yet
        # The routed gradients are complete only after flush() + join().  GradArena.sync() / FusedAdam.step() / TrainStep do
        # that explicitly; for every other caller (the reference's own ``loss.backward(); optimizer.step()`` loop,
        # runtime.py:188-189) the first launch of a backward pass registers join() as a FINAL CALLBACK of that pass: it runs on
        # the thread that called backward(), on its current stream, once the whole graph has been executed.
        self._join_queued = False

    def _view(self, p_):
        hit = self.views.get(id(p_))
        return hit[1] if (hit is not None and hit[0]() is p_) else None

    def route(self, weight, bias):
        gw = self._view(weight)
        if gw is None:
            return None
        gb = self._view(bias) if bias is not None else None
        return gw, gb

    def _kick(self):
        """hand the queued launches to the lane: after everything enqueued so far on the current stream"""
        if not self._queued:
            return
        queued, self._queued = self._queued, []
        main = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(main)
        self.stream.wait_event(ev)
        keep = []
        with torch.cuda.stream(self.stream):
            for fn, tensors, params, _ in queued:
                fn()
                keep += [t for t in tensors if t is not None]
                self._pending.append(params)
            done = torch.cuda.Event()
            done.record(self.stream)
        if self.record_streams:
            for t in keep:
                t.record_stream(self.stream)
        self._inflight.append((done, keep))
        if not torch.cuda.is_current_stream_capturing():      # (an event recorded inside a capture cannot be queried)
            while self._inflight and self._inflight[0][0].query():
                self._inflight.popleft()
        if self.batch is None or not self.batch.n:
            self.flush(kick=False)                             # nothing deferred: the gradients are complete already

    def flush(self, kick: bool = True):
        """fold every pending partial image (one launch on the lane) and report the gradients that are complete now"""
        if kick and not self.inline:
            self._kick()
        if self.batch is not None and self.batch.n:
            if self.inline:
                self.batch.run()                               # (same stream: the allocator orders any reuse after the fold)
            else:
                with torch.cuda.stream(self.stream):
                    keep = self.batch.run()
                    done = torch.cuda.Event()
                    done.record(self.stream)
                self._inflight.append((done, keep))
        pending, self._pending = self._pending, []
        if self.on_launch is not None:
            for p_ in pending:
                self.on_launch(*p_)

    def _end_of_backward(self):
        self.join()

    def _queue_join(self):
        if self._join_queued:
            return
        try:
            torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)
            self._join_queued = True
        except RuntimeError:                     # not inside a backward pass (direct calls in tests / tools): the caller joins
            pass

    def launch(self, fn, tensors, params=(None, None), gw=None):
        """Run ``fn`` on the lane after everything enqueued so far on the current stream.  The tensors it reads are kept
        ALIVE (strong references) until the lane has passed the launch: (a) the caching allocator cannot recycle them, and
        (b) a tensor with a second owner is never accumulated into IN PLACE by the autograd engine (InputBuffer::accumulate
        only steals a gradient whose use_count is 1), nor handed to a consumer as its exclusive property -- whatever the
        model code around the node does with the same gradient tensor (``a = a + b`` feeding two nodes,
        models/pwcnet_irr*.py)."""
        self._queue_join()
        if self.batch is not None and gw is not None:
            # a batch folds into each gradient at most once, and holds at most cap jobs: queued launches count
            if (self.batch.full_for(gw) or any(q[3] == gw.data_ptr() for q in self._queued)
                    or self.batch.n + len(self._queued) >= self.batch.cap - 1):
                self.flush()
        if self.inline:
            fn()
            self._pending.append(params)
            if self.batch is None or not self.batch.n:
                self.flush()
            if self.on_queue is not None and params[0] is not None:
                self.on_queue(*params)
            return
        self._queued.append((fn, tensors, params, gw.data_ptr() if gw is not None else 0))
        if len(self._queued) >= self.group:
            self._kick()
        if self.on_queue is not None and params[0] is not None:
            self.on_queue(*params)

    def join(self):
        self._join_queued = False               # (also after a backward pass that raised before its final callbacks ran)
        self.flush()
        if self.inline:
            return
        torch.cuda.current_stream().wait_stream(self.stream)
        self._inflight.clear()                   # later work on the current stream is ordered after the lane


SIDE: Optional[WgradSide] = None


def wgrad_param(x, gy, weight, bias, stride: int, dil: int, want_bias: bool = True, alpha: float = 1.0,
                acc=None):
    """Weight (+bias) gradient of one conv use.  Returns (gw, gb) tensors for autograd -- or (None, None) when the
    result was accumulated asynchronously into the gradient arena (SIDE lane).  ``acc`` = optional (gw, gb) pair to
    accumulate into (shared weights used several times inside one autograd node)."""
    routed = SIDE.route(weight, bias) if SIDE is not None else None
    if routed is not None:
        gwv, gbv = routed
        SIDE.launch(lambda: conv_wgrad(x, gy, weight.shape, stride, dil, gw=gwv, gbias=gbv if want_bias else None,
                                       alpha=alpha, defer=SIDE.batch), (x, gy),
                    (weight, bias if (want_bias and gbv is not None) else None), gw=gwv)
        return None, None
    if acc is not None:
        gw, gb = acc
    else:
        gw = None
        gb = torch.zeros(weight.shape[0], device=x.device, dtype=torch.float32) if (want_bias and bias is not None) else None
    gw = conv_wgrad(x, gy, weight.shape, stride, dil, gw=gw, gbias=gb if want_bias else None, alpha=alpha)
    return gw, gb


def lrelu_bwd_bias(gy: torch.Tensor, y: Optional[torch.Tensor], lrelu: bool, gpre: Optional[torch.Tensor],
                   gbias: Optional[torch.Tensor]) -> None:
    """gpre = gy * LeakyReLU'(y) (y = the activated output); gbias += sum over (b, h, w) of gpre."""
    B, C, H, W = gy.shape
    hip.call("irr_lrelu_bwd_bias_f32", hip.ptr(gy), hip.ptr(y) if lrelu else None, hip.ptr(gpre), hip.ptr(gbias),
             B, C, H * W, hip.bs(gy), hip.bs(y) if lrelu else 0, hip.bs(gpre) if gpre is not None else 0,
             int(lrelu), hip.stream())


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
        if res is None and alpha == 1.0:
            y = conv_forward(x, weight, bias, stride, dil, lrelu)
            act = y
        else:
            # keep the activated conv output for the LeakyReLU derivative
            act = conv_forward(x, weight, bias, stride, dil, lrelu)
            y = act * alpha if res is None else torch.add(res, act, alpha=alpha)
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
        if lrelu or (gb is not None and not bias_in_wgrad):
            gpre = torch.empty_like(g) if lrelu else None
            lrelu_bwd_bias(g, act, lrelu, gpre, gb)                  # mask and bias gradient in one HBM pass
            if lrelu:
                g = gpre
        gx = conv_dgrad(g, ctx.weight_obj, stride, dil, x.shape[2:]) if ctx.needs_input_grad[0] else None
        gw = None
        if want_w:
            if bias_in_wgrad:
                gw, gb = wgrad_param(x, g, ctx.weight_obj, ctx.bias_obj, stride, dil, want_bias=True)
            else:
                gw, _ = wgrad_param(x, g, ctx.weight_obj, None, stride, dil, want_bias=False)
        return gx, gw, gb, None, None, None, gres, None


class _CatPart(ctypes.Structure):
    """IrrCatPart of include/irr_hip.h"""
    _fields_ = [("src", ctypes.c_void_p), ("src_bs", ctypes.c_long), ("channels", ctypes.c_int), ("reserved", ctypes.c_int)]


CAT_MAX_PARTS = 8            # IRR_CAT_MAX_PARTS


def cat_channels_into(dst: torch.Tensor, parts, zero_tail: int = 0) -> None:
    """dst[:, :sum(channels)] = cat(parts, dim=1) (+ ``zero_tail`` zero channels behind them) in ONE launch
    (irr_cat_channels_f32) -- dst is a channel-slice view of the consumer's buffer.  Parts whose planes are not dense are made
    contiguous first."""
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
        hip.call("irr_cat_channels_f32", hip.ptr(view), dst.stride(0), ctypes.addressof(arr), len(chunk), B, H * W, hip.stream())
        c0 += sum(ch_ for _, _, ch_ in chunk)


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
        cat_channels_into(buf[:, 448:], parts)
        off = 448
        for i in range(5):
            co = _DenseEstimatorFn.GROW[i]
            conv_forward(buf[:, off:ctot], ws[i], bs[i], 1, 1, True, out=buf[:, off - co:off])
            off -= co
        if has_base:
            base_c = base if _planes_dense(base) else base.contiguous()
            out = conv_forward(buf[:, :ctot], ws[5], bs[5], 1, 1, False, res=base_c, alpha=1.0)
            buf[:, ctot:].copy_(out)
        else:
            out = conv_forward(buf[:, :ctot], ws[5], bs[5], 1, 1, False)
        ctx.save_for_backward(buf, *ws)
        ctx.cfg = (cin0, E, has_base, tuple(widths), int(nrelu))
        ctx.wobjs, ctx.bobjs = ws, bs
        return buf, out

    @staticmethod
    def backward(ctx, g_buf, g_out):
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
            g_est = G[:, ctot:] + g_est if g_est is not None else G[:, ctot:].clone()
        grads_w = [None] * 6
        grads_b = [None] * 6
        # conv_last first: its data gradient touches every channel (K is tiny, the launch is memory-bound) and its
        # epilogue turns the c5 slice into a pre-activation gradient.  Then the buffer is back-propagated COLUMN-WISE:
        # for each slice T = c4, c3, c2, c1, x (in that order) ONE launch sums the contributions of all later layers,
        # reading their concatenated pre-activation gradients G[:, :t0] (contiguous by construction) against a
        # combined packed weight matrix, accumulates into G[:, T] once and applies LeakyReLU'(buf[:, T]) in the same
        # epilogue.  Versus layer-by-layer accumulation this replaces up to five small-K read-modify-write launches
        # per slice by a single large-K one.  Bias gradients ride on the wgrad launches.
        if g_est is not None:
            grads_w[5], grads_b[5] = wgrad_param(buf[:, :ctot], g_est, ctx.wobjs[5], ctx.bobjs[5], 1, 1)
            conv_dgrad(g_est, ws[5], 1, 1, (H, W), gx=G[:, :ctot], accumulate=True, mask=buf[:, :ctot], nmask=32)
        else:
            lrelu_bwd_bias(G[:, :32], buf[:, :32], True, G[:, :32], None)
        use_x3 = [bool(x3_code(B, t0, H, W, t1 - t0, 3, 1, 1)) for (t0, t1) in ((32, 96), (96, 192), (192, 320), (320, 448), (448, ctot))]
        packs = _dense_column_packs(ctx.wobjs[:5], cin0, tuple(use_x3))
        grads_w[4], grads_b[4] = wgrad_param(buf[:, 32:ctot], G[:, :32], ctx.wobjs[4], ctx.bobjs[4], 1, 1)   # conv5
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
            LAUNCHES["dense_column_x3" if use_x3[k_] else "dense_column_f32"] += 1
            if use_x3[k_]:
                args = ("irr_conv2d_fwd_x3", hip.ptr(G), hip.ptr(packs[k_]), None, None, hip.ptr(G[:, t0:t1]), B, t0, H, W,
                        t1 - t0, 1, hip.bs(G), hip.bs(G), 0, 0, 1.0, 1, *margs, hip.stream())
                variant = 100000 + x3_code(B, t0, H, W, t1 - t0, 3, 1, 1)
            else:
                args = ("irr_conv2d_fwd_f32", hip.ptr(G), hip.ptr(packs[k_]), None, None, hip.ptr(G[:, t0:t1]), B, t0, H, W,
                        t1 - t0, H, W, 3, 1, 1, hip.bs(G), hip.bs(G), 0, 0, 1.0, 1, *margs, hip.stream())
                variant = hip.lib().irr_conv2d_fwd_variant(B, t1 - t0, H, W, 3)
            if TIMER is None:
                _call_conv(args)
            else:
                TIMER.wrap(variant, 2.0 * B * H * W * t0 * (t1 - t0) * 9, lambda: _call_conv(args), "dgrad")
            if not last:                                   # G[:, t0:t1] is now the pre-activation gradient of conv(4-k_)
                i = 3 - k_
                grads_w[i], grads_b[i] = wgrad_param(buf[:, t1:ctot], G[:, t0:t1], ctx.wobjs[i], ctx.bobjs[i], 1, 1)
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


def _dense_column_packs(ws5, cin0: int, use_x3=(False,) * 5):
    """Combined (transposed, flipped) packed weights for the five column targets c4, c3, c2, c1, x of the DenseNet
    buffer; cached on the first weight tensor (per kernel-family choice) and rebuilt when any of the five conv weights
    changed -- as sub-jobs of the batched pack launch once they are registered.  The buffers are allocated (zeroed) once:
    rows and columns that no layer covers stay zero, the sub-jobs only rewrite what they own.
    use_x3[k]: column k runs on irr_conv2d_fwd_x3 and needs the pre-split layout."""
    def cur_tags():
        return tuple((w.data_ptr(), w._version) for w in ws5) + (WEIGHT_EPOCH[0], cin0)
    tags = cur_tags()
    holder = ws5[0].__dict__.setdefault("_irr_dense_packs", {}).setdefault((tuple(use_x3), cin0), {})
    if holder.get("tag") == tags:
        return holder["packs"]
    reg = _registry(ws5[0].device)
    if "packs" in holder:
        old = holder.get("tag")
        if old is not None and old[-2] == tags[-2] and reg.epoch == WEIGHT_EPOCH[0] and old[:-2] != tags[:-2]:
            WEIGHT_EPOCH[0] += 1                             # rewritten behind the caches' back (see _announce_rewrite)
        if reg.refresh() and holder.get("tag") == cur_tags():
            return holder["packs"]
        tags = cur_tags()
    in0 = [448, 320, 192, 96, 32]                         # first buffer channel read by conv1..conv5
    row0 = {5: 0, 4: 32, 3: 96, 2: 192, 1: 320}           # row (= G channel) where conv i's gradient slice starts
    bounds = [(32, 96), (96, 192), (192, 320), (320, 448), (448, 448 + cin0)]
    dev = ws5[0].device
    fresh = "packs" not in holder
    packs = [] if fresh else holder["packs"]
    lib = hip.lib()
    wrefs = [weakref.ref(w) for w in ws5]

    def retag(holder=holder, wrefs=wrefs):
        live = [r() for r in wrefs]
        if all(w is not None for w in live):
            holder["tag"] = tuple((w.data_ptr(), w._version) for w in live) + (WEIGHT_EPOCH[0], cin0)

    for k_, (t0, t1) in enumerate(bounds):
        n = t1 - t0
        cop = (n + 31) // 32 * 32
        if fresh:
            if use_x3[k_]:
                packs.append(torch.zeros(lib.irr_conv_x3_packed_bytes(t0, n), device=dev, dtype=torch.uint8))
            else:
                packs.append(torch.zeros(lib.irr_conv_packed_weight_elems(t0, n, 3), device=dev, dtype=torch.float32))
        wp = packs[k_]
        for i in (5, 4, 3, 2, 1):
            if in0[i - 1] > t0:
                continue                                  # conv i does not read this slice
            wsrc = ws5[i - 1]
            w = wsrc.detach().contiguous()
            wcin, wcout, c0 = w.shape[1], w.shape[0], t0 - in0[i - 1]
            if use_x3[k_]:
                hip.call("irr_conv_pack_weights_x3_sub", hip.ptr(w), hip.ptr(wp), wcin, wcout, t0, c0, n, row0[i], hip.stream())
                builder = (lambda job, wptr, wp=wp, a=(wcin, wcout, t0, c0, n, row0[i]):
                           lib.irr_conv_pack_job_x3_sub(job, wptr, wp.data_ptr(), *a))
            else:
                hip.call("irr_conv_pack_weights_sub_f32", hip.ptr(w), hip.ptr(wp), wcin, wcout, 3, c0, n, cop, row0[i], hip.stream())
                builder = (lambda job, wptr, wp=wp, a=(wcin, wcout, 3, c0, n, cop, row0[i]):
                           lib.irr_conv_pack_job_sub_f32(job, wptr, wp.data_ptr(), *a))
            LAUNCHES["pack_single"] += 1
            if wsrc.is_contiguous():
                reg.register((id(ws5[0]), "dense", tuple(use_x3), cin0, k_, i), wsrc, wp, builder, retag)
    holder["tag"] = tags
    holder["packs"] = packs
    return packs


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
    return _DenseEstimatorFn.apply(len(parts), base, int(preact_grad_channels), *parts, *weights_and_biases)


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
    def forward(ctx, x, res, cfg, *wb):
        ws, bs = wb[0::2], wb[1::2]
        x = x if _planes_dense(x) else x.contiguous()
        acts = []
        cur = x
        n = len(ws)
        for i in range(n):
            stride, dil, lrelu = cfg[i]
            last = i == n - 1
            if last and res is not None:
                res_c = res if _planes_dense(res) else res.contiguous()
                if lrelu:
                    a = conv_forward(cur, ws[i], bs[i], stride, dil, True)      # keep the activation for its mask
                    acts.append(a)
                    cur = torch.add(res_c, a)
                else:
                    cur = conv_forward(cur, ws[i], bs[i], stride, dil, False, res=res_c)
                    acts.append(None)
            else:
                cur = conv_forward(cur, ws[i], bs[i], stride, dil, lrelu)
                acts.append(cur)
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
        if cfg[-1][2]:                                        # activation on the chain output: one pass on a small tensor
            gpre = torch.empty_like(gy)
            lrelu_bwd_bias(gy, a_last, True, gpre, None)
            g = gpre
        grads = [None] * (2 * n)
        for i in range(n - 1, -1, -1):
            stride, dil, _ = cfg[i]
            inp = acts[i - 1] if i > 0 else x
            grads[2 * i], grads[2 * i + 1] = wgrad_param(inp, g, ws[i], ctx.bias_objs[i], stride, dil)
            if i > 0:
                prev_lrelu = cfg[i - 1][2]
                g = conv_dgrad(g, ws[i], stride, dil, inp.shape[2:], mask=inp if prev_lrelu else None,
                               nmask=inp.shape[1] if prev_lrelu else 0)
            elif ctx.needs_input_grad[0]:
                g = conv_dgrad(g, ws[i], stride, dil, inp.shape[2:])
            else:
                g = None
        return (g, gres, None) + tuple(grads)


def conv_chain(x, layers, res=None):
    """layers: sequence of modules exposing .weight, .bias, .stride, .dilation, .is_relu (modules.ConvBlock)."""
    if not x.is_cuda:
        raise RuntimeError("irr_amd conv runs on the HIP device only (no CPU fallback)")
    cfg = tuple((int(l.stride), int(l.dilation), bool(l.is_relu)) for l in layers)
    wb = []
    for l in layers:
        wb += [l.weight, l.bias]
    return _ConvChainFn.apply(x, res, cfg, *wb)


# ----------------------------------------------------------------------------------------------
# autograd: OccUpsampleNetwork (models/irr_modules.py:30-56) as ONE node
# ----------------------------------------------------------------------------------------------
def _padded_cin(weight: torch.Tensor, cpad: int) -> torch.Tensor:
    """persistent copy of ``weight`` (Cout, Cin, k, k) with its input channels zero-padded to ``cpad`` -- refreshed when the
    parameter changed (same tag as the packed-weight caches); its own packed copies follow through its version counter"""
    holder = weight.__dict__.setdefault("_irr_cinpad", {})
    w = weight.detach()
    tag = _weight_tag(w)
    hit = holder.get(cpad)
    if hit is not None and hit[0] == tag:
        return hit[1]
    wp = hit[1] if hit is not None else torch.zeros(w.shape[0], cpad, w.shape[2], w.shape[3], device=w.device, dtype=torch.float32)
    # a DERIVED tensor: rewriting it here is a consequence of a parameter update that has been noticed already, not a new one
    # (_announce_rewrite would move the weight epoch again and every later call would find its tag stale once more)
    wp.__dict__["_irr_derived"] = True
    wp[:, :w.shape[1]].copy_(w)
    holder[cpad] = (tag, wp)
    return wp


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
        cat_channels_into(x_in, (occ_up,) + tuple(parts[1:]), zero_tail=cpad - cin)
        w_first = _padded_cin(w_init, cpad) if cpad > cin else w_init
        x_init = conv_forward(x_in, w_first, b_init, 1, 1, True, real_cin=cin)
        xs = [x_init]
        ts = []
        for _ in range(3):
            t = conv_forward(xs[-1], w_r0, b_r0, 1, 1, True)
            ts.append(t)
            xs.append(conv_forward(t, w_r1, b_r1, 1, 1, False, res=xs[-1], alpha=mul_const))
        e, x2 = conv_forward_skip(xs[-1], w_end, b_end, True, x_init)
        o = conv_forward(x2, w_out, b_out, 1, 1, True)
        out = torch.add(o, occ_up)
        ctx.mul_const = mul_const
        ctx.widths = widths
        ctx.wobjs = (w_init, w_r0, w_r1, w_end, w_out)
        ctx.bobjs = (b_init, b_r0, b_r1, b_end, b_out)
        ctx.save_for_backward(x_in, xs[0], xs[1], xs[2], xs[3], ts[0], ts[1], ts[2], e, x2, o)
        return out

    @staticmethod
    def backward(ctx, g_out):
        x_in, x0, x1, x2r, x3, t1, t2, t3, e, x2, o = ctx.saved_tensors
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
        # out = occ_up + lrelu(conv_out(x2))
        gpre_o = torch.empty_like(g_out)
        gb_out = z(w_out.shape[0])
        lrelu_bwd_bias(g_out, o, True, gpre_o, gb_out)                       # 1-channel tensor
        gw_out, _ = wgrad_param(x2, gpre_o, w_out, None, 1, 1, want_bias=False)
        # x2 = x_init + e, e = lrelu(conv_end(x3)): the gradient of x2 is needed raw (g_x2: the skip into x_init) and multiplied by
        # LeakyReLU'(e) (gpre_e: into res_end_conv).  Both come out of the out_convs data-gradient launch where its quad kernel
        # applies (one pass less over two 32-channel full-resolution maps); the bias gradient then rides on the wgrad launch.
        B_, _, H_, W_ = x2.shape
        dual = (w_out.shape[0] == 1 and W_ % 4 == 0 and not os.environ.get("IRR_OCCUP_NO_DUAL_DGRAD"))       # (A/B switch)
        if dual:
            g_x2 = torch.empty(B_, w_out.shape[1], H_, W_, device=dev, dtype=torch.float32)
            gpre_e = torch.empty_like(g_x2)
            LAUNCHES["dgrad_smallco"] += 1
            hip.call("irr_conv2d_smallco_dgrad_dual_f32", hip.ptr(gpre_o), hip.ptr(w_out.detach().contiguous()), hip.ptr(gpre_e),
                     hip.ptr(g_x2), hip.ptr(e), B_, w_out.shape[1], H_, W_, 1, hip.bs(gpre_o), hip.bs(gpre_e), hip.bs(g_x2), hip.bs(e),
                     hip.stream())
            gw_end, gb_end = wgrad_param(x3, gpre_e, w_end, b_end, 1, 1, want_bias=True)
        else:
            g_x2 = conv_dgrad(gpre_o, w_out, 1, 1, hw_)                      # (B,32,H,W); also the gradient of x_init via the skip
            gpre_e = torch.empty_like(g_x2)
            gb_end = z(w_end.shape[0])
            lrelu_bwd_bias(g_x2, e, True, gpre_e, gb_end)
            gw_end, _ = wgrad_param(x3, gpre_e, w_end, None, 1, 1, want_bias=False)
        g_x = conv_dgrad(gpre_e, w_end, 1, 1, hw_)                           # gradient w.r.t. x3
        # three residual blocks with shared weights: x_i = x_{i-1} + mc * conv_r1(t_i), t_i = lrelu(conv_r0(x_{i-1}))
        routed = SIDE is not None and SIDE.route(w_r0, b_r0) is not None
        acc_r0 = None if routed else (torch.zeros_like(w_r0), z(w_r0.shape[0]))
        acc_r1 = None if routed else (torch.zeros_like(w_r1), z(w_r1.shape[0]))
        xs = [x0, x1, x2r]
        ts = [t1, t2, t3]
        for i in (2, 1, 0):
            wgrad_param(ts[i], g_x, w_r1, b_r1, 1, 1, alpha=mc, acc=acc_r1)
            gpre_t = conv_dgrad(g_x, w_r1, 1, 1, hw_, mask=ts[i], nmask=ts[i].shape[1], alpha=mc)
            wgrad_param(xs[i], gpre_t, w_r0, b_r0, 1, 1, acc=acc_r0)
            if i > 0:
                g_x = conv_dgrad(gpre_t, w_r0, 1, 1, hw_, res=g_x)                # skip + branch in one launch
            else:
                # x_0 = x_init: add the x2 skip gradient (accumulate into g_x2) and apply init_conv's LeakyReLU'
                conv_dgrad(gpre_t, w_r0, 1, 1, hw_, gx=g_x2, accumulate=True, res=g_x, mask=x0, nmask=x0.shape[1])
        gw_r0, gb_r0 = acc_r0 if acc_r0 is not None else (None, None)
        gw_r1, gb_r1 = acc_r1 if acc_r1 is not None else (None, None)
        gpre_init = g_x2
        x_real = x_in[:, :cin] if x_in.shape[1] > cin else x_in
        gw_init, gb_init = wgrad_param(x_real, gpre_init, w_init, b_init, 1, 1)
        gparts = [None] * nparts
        if any(ctx.needs_input_grad[2:2 + nparts]):
            w_first = _padded_cin(w_init, x_in.shape[1]) if x_in.shape[1] > cin else w_init
            g_xin = conv_dgrad(gpre_init, w_first, 1, 1, hw_, real_cin=cin)
            c0 = 0
            for i, wd in enumerate(widths):
                if ctx.needs_input_grad[2 + i]:
                    gparts[i] = g_xin[:, c0:c0 + wd]
                c0 += wd
        if ctx.needs_input_grad[2]:                           # occ_up: channel 0 of the input AND the final skip
            gparts[0] = g_out + gparts[0] if gparts[0] is not None else g_out
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
