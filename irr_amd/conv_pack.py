"""Packed-weight caches of the conv kernels (irr_amd.conv): every packed copy of a live conv weight -- fp32 MFMA layout, transposed +
flipped (data gradient), combined DenseNet column matrices, bf16x3 pre-split -- is cached ON the parameter tensor, registered as a job
of the batched pack launch (csrc/pack_batch.hip) the first time it is built, and refreshed with ONE launch after an optimizer step.
"""
from __future__ import annotations

import collections
import ctypes
import weakref

import torch

from . import hip

# launches per kernel family since the last clear() -- what a step was ROUTED to (tests assert on it, bench.py reports it)
LAUNCHES: "collections.Counter[str]" = collections.Counter()

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
        self.amax_sources = {}          # id(weight) -> (weakref(weight), its one-element max |w| tensor): the h2 packs' scales
        self.amax_groups = []           # (weakrefs of the weights of a combined matrix, its one-element group maximum)

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
        self.refresh_amax()
        with hip.device_of(table):
            hip.call("irr_conv_pack_batch", hip.ptr(table), njobs, nblocks, hip.stream())
        LAUNCHES["pack_batch"] += 1
        for _, e, _ in live:
            e[3]()
        return True


    def pin(self):
        """strong references to everything the batched repack launch touches right now: every live registered weight, its packed copy,
        the job table, the weights' maximum slots.  A CAPTURED step replays that launch with the pointers of capture time -- including
        the packs of any OTHER model alive in the process then (this registry is per device, not per model): were such a model freed
        later, every replay would write its packs into memory that belongs to somebody else by now.  irr_amd.train.GraphedTrainStep
        holds the result for as long as its graph lives (and takes it BEFORE the capture starts, so that no entry can disappear -- and
        the job table be rebuilt with a host-to-device copy -- in the middle of it)."""
        held = []
        for _, (ref, dst, _b, _r) in list(self.entries.items()):
            w = ref()
            if w is not None:
                held.append((w, dst))
        for _, (ref, t) in list(self.amax_sources.items()):
            w = ref()
            if w is not None:
                held.append((w, t))
        for refs, g in self.amax_groups:
            held.append(([r() for r in refs], g))
        held.append(self._table)
        return held

    def refresh_amax(self) -> None:
        """max |w| of every weight with an h2 pack (one multi-tensor launch + one copy per weight's slot) and the group maxima of
        the combined matrices -- before the batched repack reads them"""
        live = []
        for key, (ref, t) in list(self.amax_sources.items()):
            w = ref()
            if w is None:
                del self.amax_sources[key]
            else:
                live.append((w.detach(), t))
        if not live:
            return
        norms = torch._foreach_norm([w for w, _ in live], float("inf"))
        torch._foreach_copy_([t for _, t in live], [n_.reshape(1) for n_ in norms])
        LAUNCHES["amax_weights"] += 1
        alive = []
        for refs, g in self.amax_groups:
            ws = [r() for r in refs]
            if all(w is not None for w in ws):
                torch.amax(torch.cat([w.__dict__["_irr_amax"] for w in ws]).reshape(1, -1), dim=1, out=g)
                alive.append((refs, g))
        self.amax_groups = alive                              # (groups of dead models go)


_REGISTRIES = {}


def pin_all():
    """_PackRegistry.pin() of every device's registry"""
    return [r.pin() for r in _REGISTRIES.values()]


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


def _packed(weight: torch.Tensor, transpose: bool, slot: str, nbytes_fn, dtype, single, builder_name, amax=None):
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
        elif builder_name == "irr_conv_pack_job_h2":
            at = amax(weight)                                  # (the one-element tensor, not the parameter: entries hold weak references)
            builder = lambda job, wptr, wp=wp, at=at: fn(job, wptr, wp.data_ptr(), lcin, lcout, int(transpose), at.data_ptr())
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

def packed_weights_x3(weight: torch.Tensor, transpose: bool) -> torch.Tensor:
    """Pre-split (3 x bf16) packed copy of ``weight`` for irr_conv2d_fwd_x3 (see _packed)."""
    assert weight.shape[2] == 3
    return _packed(weight, transpose, "_irr_packed_x3", lambda ci, co, k: hip.lib().irr_conv_x3_packed_bytes(ci, co), torch.uint8,
                   lambda wc, wp, ci, co, k, tr: hip.call("irr_conv_pack_weights_x3", hip.ptr(wc), hip.ptr(wp), ci, co, tr,
                                                          hip.stream()),
                   "irr_conv_pack_job_x3")



# ---- fp16x2 ("h2") packs: one power-of-two scale per packed matrix, from max |w| -----------------------------------------------
# Every weight that has an h2 pack owns a persistent one-element device tensor with its max |w| (cached on the parameter).  The
# registry refreshes ALL of them with one multi-tensor launch right before the batched repack (_PackRegistry.refresh), so the
# pack jobs -- which keep the tensors' addresses -- always read current values.
def _weight_amax(weight: torch.Tensor) -> torch.Tensor:
    t = weight.__dict__.get("_irr_amax")
    if t is None or t.device != weight.device:
        t = weight.__dict__["_irr_amax"] = torch.zeros(1, device=weight.device, dtype=torch.float32)
        _registry(weight.device).amax_sources[id(weight)] = (weakref.ref(weight), t)
    return t


def _refresh_weight_amax(weight: torch.Tensor) -> torch.Tensor:
    t = _weight_amax(weight)
    torch.amax(weight.detach().abs().reshape(1, -1), dim=1, out=t)
    return t


def packed_weights_h2(weight: torch.Tensor, transpose: bool) -> torch.Tensor:
    """Pre-split (2 x fp16 of w * 2^e) packed copy of ``weight`` for irr_conv2d_fwd_h2 (see _packed)"""
    assert weight.shape[2] == 3
    return _packed(weight, transpose, "_irr_packed_h2", lambda ci, co, k: hip.lib().irr_conv_h2_packed_bytes(ci, co), torch.uint8,
                   lambda wc, wp, ci, co, k, tr: hip.call("irr_conv_pack_weights_h2", hip.ptr(wc), hip.ptr(wp), ci, co, tr,
                                                          hip.ptr(_refresh_weight_amax(weight)), hip.stream()),
                   "irr_conv_pack_job_h2", amax=_weight_amax)


def _dense_column_packs(ws5, cin0: int, use_x3=(False,) * 5):
    """Combined (transposed, flipped) packed weights for the five column targets c4, c3, c2, c1, x of the DenseNet
    buffer; cached on the first weight tensor (per kernel-family choice) and rebuilt when any of the five conv weights
    changed -- as sub-jobs of the batched pack launch once they are registered.  The buffers are allocated (zeroed) once:
    rows and columns that no layer covers stay zero, the sub-jobs only rewrite what they own.
    use_x3[k]: 0 = column k runs on the fp32 kernel, 1 = on irr_conv2d_fwd_x3 (bf16x3 pre-split layout), 2 = on irr_conv2d_fwd_h2
    (fp16x2 layout; the five matrices share ONE scale, the maximum over the block's five weights)."""
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
    gamax = None
    if any(u == 2 for u in use_x3):
        gamax = holder.get("gamax")
        if gamax is None:
            gamax = holder["gamax"] = torch.zeros(1, device=dev, dtype=torch.float32)
            reg.amax_groups.append((wrefs, gamax))
        torch.amax(torch.cat([_refresh_weight_amax(w) for w in ws5]).reshape(1, -1), dim=1, out=gamax)

    def retag(holder=holder, wrefs=wrefs):
        live = [r() for r in wrefs]
        if all(w is not None for w in live):
            holder["tag"] = tuple((w.data_ptr(), w._version) for w in live) + (WEIGHT_EPOCH[0], cin0)

    for k_, (t0, t1) in enumerate(bounds):
        n = t1 - t0
        cop = (n + 31) // 32 * 32
        if fresh:
            if use_x3[k_] == 2:
                packs.append(torch.zeros(lib.irr_conv_h2_packed_bytes(t0, n), device=dev, dtype=torch.uint8))
            elif use_x3[k_]:
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
            if use_x3[k_] == 2:
                hip.call("irr_conv_pack_weights_h2_sub", hip.ptr(w), hip.ptr(wp), wcin, wcout, t0, c0, n, row0[i], hip.ptr(gamax), hip.stream())
                builder = (lambda job, wptr, wp=wp, a=(wcin, wcout, t0, c0, n, row0[i]), g=gamax:
                           lib.irr_conv_pack_job_h2_sub(job, wptr, wp.data_ptr(), *a, g.data_ptr()))
            elif use_x3[k_]:
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


