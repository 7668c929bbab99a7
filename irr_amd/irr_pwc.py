"""IRR-PWC on MI355X -- drop-in for the reference's ``models.IRR_PWC`` (= models/IRR_PWC.py ``PWCNet``).

Same constructor (``PWCNet(args, div_flow=0.05)``), same ``forward(input_dict) -> dict`` contract
(train: ``{'flow': 7 lists, 'occ': 7 lists}``; eval: full-resolution ``{'flow', 'occ'}``), same 124
``state_dict`` keys and the same attributes (``_div_flow, search_range, num_chs, output_level, num_levels,
corr_params``).  What differs is the execution plan:

* both flow directions are processed as ONE batch of 2B samples (``[x1; x2]`` against ``[x2; x1]``): the
  decoders share their weights across directions (models/IRR_PWC.py:32-46), so this halves the number of
  launches and doubles the pixel count every MFMA conv tile sees; results are identical because every
  operator is per-sample;
* the reference's ``rescale_flow`` aliasing (models/pwc_modules.py:70-82, observable at
  models/IRR_PWC.py:128-138) is written out explicitly, alias-free: G = S*flow_cont is what the image
  warp and RefineFlow see, the level's ``flow_cont`` output is S*G, and the refined flow is scaled once;
* every tensor operation is a kernel of libirr_hip.so (cost volume, warp+mask, bilinear / nearest
  resize, MFMA convs, bilateral refinement tail).
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import functional as Fn
from . import harness as _harness
from .modules import (ContextNetwork, FeatureExtractor, FlowEstimatorDense, OccContextNetwork, OccEstimatorDense,
                      OccUpsampleNetwork, RefineFlow, RefineOcc, WarpingLayer, conv, initialize_msra)


class _SwapHalves(torch.autograd.Function):
    """[a; b] -> [b; a] along the batch axis.  As one node its backward is ONE copy kernel (the same swap of the gradient)
    instead of two zero-filled slice gradients, two copies and an add."""

    @staticmethod
    def forward(ctx, t):
        b = t.shape[0] // 2
        return torch.cat([t[b:], t[:b]], dim=0)

    @staticmethod
    def backward(ctx, g):
        b = g.shape[0] // 2
        return torch.cat([g[b:], g[:b]], dim=0)


def _swap_halves(t: torch.Tensor) -> torch.Tensor:
    return _SwapHalves.apply(t) if t.requires_grad else torch.cat([t[t.shape[0] // 2:], t[:t.shape[0] // 2]], dim=0)


class _SplitHalves(torch.autograd.Function):
    """[a; b] -> (a, b) (views).  Backward: one concatenation of the two gradients (zeros for an unused half) instead of two
    zero-filled full-size slice gradients and an add."""

    @staticmethod
    def forward(ctx, t):
        b = t.shape[0] // 2
        ctx.half = (b,) + tuple(t.shape[1:])
        return t[:b], t[b:]

    @staticmethod
    def backward(ctx, g1, g2):
        if g1 is None and g2 is None:
            return None
        if g1 is not None and g2 is not None and g1._base is not None and g1._base is g2._base:
            base = g1._base                               # the two halves of one buffer (losses._paired_grads): no copy
            if base.is_contiguous() and base.shape == (2 * ctx.half[0],) + ctx.half[1:] and g1.data_ptr() == base.data_ptr() \
                    and g2.data_ptr() == base.data_ptr() + g1.numel() * g1.element_size() and g1.is_contiguous() and g2.is_contiguous():
                return base
        ref = g1 if g1 is not None else g2
        z = None
        if g1 is None or g2 is None:
            z = torch.zeros(ctx.half, device=ref.device, dtype=ref.dtype)
        return torch.cat([g1 if g1 is not None else z, g2 if g2 is not None else z], dim=0)


def _split_halves(t: torch.Tensor):
    if t.requires_grad:
        return _SplitHalves.apply(t)
    b = t.shape[0] // 2
    return t[:b], t[b:]


_CORR_PREMASK = os.environ.get("IRR_CORR_NO_PREMASK") is None     # A/B switch: the cost-volume gradient kernels apply LeakyReLU' themselves


class PWCNet(nn.Module):
    def __init__(self, args, div_flow=0.05, mask_threshold: float = 1.0):
        super().__init__()
        self.args = args
        self._div_flow = div_flow
        self.search_range = 4
        self.num_chs = [3, 16, 32, 64, 96, 128, 196]
        self.output_level = 4
        self.num_levels = 7
        self.leakyRELU = nn.LeakyReLU(0.1, inplace=True)

        self.feature_pyramid_extractor = FeatureExtractor(self.num_chs)
        self.warping_layer = WarpingLayer(mask_threshold)

        self.dim_corr = (self.search_range * 2 + 1) ** 2
        self.num_ch_in_flo = self.dim_corr + 32 + 2
        self.num_ch_in_occ = self.dim_corr + 32 + 1

        self.flow_estimators = FlowEstimatorDense(self.num_ch_in_flo)
        self.context_networks = ContextNetwork(self.num_ch_in_flo + 448 + 2)
        self.occ_estimators = OccEstimatorDense(self.num_ch_in_occ)
        self.occ_context_networks = OccContextNetwork(self.num_ch_in_occ + 448 + 1)
        self.occ_shuffle_upsample = OccUpsampleNetwork(11, 1)

        self.conv_1x1 = nn.ModuleList([conv(196, 32, kernel_size=1, stride=1, dilation=1),
                                       conv(128, 32, kernel_size=1, stride=1, dilation=1),
                                       conv(96, 32, kernel_size=1, stride=1, dilation=1),
                                       conv(64, 32, kernel_size=1, stride=1, dilation=1)])
        self.conv_1x1_1 = conv(16, 3, kernel_size=1, stride=1, dilation=1)

        self.refine_flow = RefineFlow(2 + 1 + 32)
        self.refine_occ = RefineOcc(1 + 32 + 32)
        self.corr_params = {"pad_size": self.search_range, "kernel_size": 1, "max_disp": self.search_range,
                            "stride1": 1, "stride2": 1, "corr_multiply": 1}
        initialize_msra(self.modules())
        # The occlusion decoder + context network of the coarse levels on a second HIP stream: their launches are too small to fill the
        # chip and independent of the flow branch (profiles/r5_branch_pairing_bound.txt: 151.6 -> 150.0 ms per step with levels < 4,
        # 150.5 with < 3, nothing more with < 5; round 1 had measured -1 % with all levels).  ON since the end of round 5: until the
        # store hazard of conv_x3s_kernel was fixed this configuration failed the bs32 oracle test.
        # IRR_BRANCH_STREAMS=0: one stream.
        self.branch_streams = os.environ.get("IRR_BRANCH_STREAMS", "1") != "0"
        self.branch_levels = int(os.environ.get("IRR_BRANCH_LEVELS", "4"))     # pyramid levels l < this use the second stream

    # the validity-mask threshold of WarpingLayer: 1.0 = reference as-is, 0.9999 = robust parity mode
    @property
    def mask_threshold(self) -> float:
        return self.warping_layer.mask_threshold

    @mask_threshold.setter
    def mask_threshold(self, v: float) -> None:
        self.warping_layer.mask_threshold = float(v)

    def _scale_tensors(self, s_loc, s_glb, dev):
        """(1,2,1,1) device tensors of the per-level flow scales, uploaded once per (scales, device)"""
        cache = self.__dict__.setdefault("_scale_cache", {})
        key = (s_loc, s_glb, str(dev))
        if key not in cache:
            cache[key] = (torch.tensor(s_loc, device=dev, dtype=torch.float32).view(1, 2, 1, 1),
                          torch.tensor(s_glb, device=dev, dtype=torch.float32).view(1, 2, 1, 1))
        return cache[key]

    def _branch_stream(self, dev, level=0):
        """second HIP stream for the occlusion branch of the coarse levels (IRR_BRANCH_STREAMS=0: one stream)"""
        if not self.branch_streams or dev.type != "cuda" or level >= self.branch_levels:
            return None
        st = self.__dict__.get("_side_stream")
        if st is None or st.device != dev:
            st = torch.cuda.Stream(device=dev)
            self.__dict__["_side_stream"] = st
            # The shared occlusion decoders are used on this stream at the coarse levels and on the caller's stream at level 4, so a
            # parameter's AccumulateGrad node (bound to the stream of its FIRST use) receives gradients from both: the engine orders them
            # with a device-side event wait between the two streams (no host synchronisation; needed, not "unnecessary") and torch
            # warns about it once per process.  The mismatch is intentional (profiles/NOTES.md E.8).
            quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
            if quiet is not None:
                quiet(False)
        return st

    def forward(self, input_dict):
        x1_raw, x2_raw = input_dict['input1'], input_dict['input2']
        B, _, H, W = x1_raw.shape
        _harness.auto_install(self)            # no-op unless a foreign training loop drives the model (irr_amd/harness.py)
        div = self._div_flow
        dev = x1_raw.device

        raw = torch.cat([x1_raw, x2_raw], dim=0)                 # [x1; x2]   (2B)
        pyr = self.feature_pyramid_extractor(raw) + [raw]        # coarsest first; both images in one pass

        def warp_other(x, fl):
            """warp the OTHER image's tensor by this direction's flow: sample b of [x1; x2] reads sample (b + B) % 2B"""
            return self.warping_layer(x, fl, H, W, div, swap_halves=True)

        h0, w0 = pyr[0].shape[2:]
        flow = torch.zeros(2 * B, 2, h0, w0, device=dev)          # [flow_f; flow_b]
        occ = torch.zeros(2 * B, 1, h0, w0, device=dev)           # [occ_f ; occ_b ]
        flows, occs = [], []
        # the raw images at every refinement level (models/IRR_PWC.py:126-127): one autograd node for the five resizes -- when the inputs
        # require grad (the reference's _step marks them, runtime.py:158-162) their five sparse full-size gradients meet in ONE buffer
        # instead of being added pairwise by the engine (IRR_NO_RESIZE_MULTI=1: A/B switch)
        n_ref = min(self.output_level + 1, len(pyr))
        imgs = (Fn.resize_bilinear_ac_multi(raw, [tuple(pyr[l_].shape[2:]) for l_ in range(n_ref)])
                if not os.environ.get("IRR_NO_RESIZE_MULTI") else None)

        for l, x in enumerate(pyr):
            h, w = x.shape[2:]
            if l <= self.output_level:
                if l == 0:
                    xo_warp = _swap_halves(x)                     # the other image's features (6x7: the only swapped copy)
                else:
                    flow = Fn.resize_bilinear_ac(flow, h, w)
                    occ = Fn.resize_bilinear_ac(occ, h, w)
                    xo_warp = warp_other(x, flow)
                # cost volume + LeakyReLU fused; its only consumers are the two estimators below, which hand back the gradient
                # already multiplied by LeakyReLU' (PRE = 81 channels): the gradient kernels never read the 81-plane output
                corr = Fn.cost_volume(x, xo_warp, lrelu=True, grad_is_preactivation=_CORR_PREMASK)

                x_1by1 = self.conv_1x1[l](x) if l != self.output_level else x

                s_loc = (float(w / W / div), float(h / H / div))      # to_local  (pwc_modules.py:72-73)
                s_glb = (float(W * div / w), float(H * div / h))      # to_global (pwc_modules.py:75-76)
                t_loc, t_glb = self._scale_tensors(s_loc, s_glb, dev)
                flow = flow * t_loc

                # estimator + "est = flow + res" + cat([x_intm, est]) in one cat-free node (conv.dense_estimator)
                ctx_in, flow_est = self.flow_estimators.forward_residual((corr, x_1by1, flow), flow, self.dim_corr if _CORR_PREMASK else 0)
                flow_cont = self.context_networks(ctx_in, res=flow_est)

                # The occlusion decoder + context network are independent of the flow branch until refine_occ: at the coarse levels
                # they run on a second HIP stream (autograd replays their backward there too; IRR_BRANCH_STREAMS=0: one stream).
                occ_in = (corr, x_1by1, occ)
                side = self._branch_stream(dev, l)
                if side is not None:
                    main = torch.cuda.current_stream()
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        ctx_in_o, occ_est = self.occ_estimators.forward_residual(occ_in, occ, self.dim_corr if _CORR_PREMASK else 0)
                        occ_cont = self.occ_context_networks(ctx_in_o, res=occ_est)
                    for t_ in occ_in:
                        t_.record_stream(side)
                    occ.record_stream(side)
                    pending_join = (main, side, occ_cont)
                else:
                    ctx_in_o, occ_est = self.occ_estimators.forward_residual(occ_in, occ, self.dim_corr if _CORR_PREMASK else 0)
                    occ_cont = self.occ_context_networks(ctx_in_o, res=occ_est)
                    pending_join = None

                # refinement (models/IRR_PWC.py:126-138, alias-free)
                img = imgs[l] if imgs is not None else Fn.resize_bilinear_ac(raw, h, w)
                G = flow_cont * t_glb
                img_o_warp = warp_other(img, G)
                flow = self.refine_flow(G.detach(), img - img_o_warp, x_1by1, scale=s_glb)   # incl. to_global
                flow_cont = G * t_glb

                x_1by1_o_warp = warp_other(x_1by1, flow)
                if pending_join is not None:
                    pending_join[0].wait_stream(pending_join[1])
                    occ_cont.record_stream(pending_join[0])
                occ = self.refine_occ(occ_cont.detach(), x_1by1, x_1by1 - x_1by1_o_warp)

                flows.append([*_split_halves(flow_cont), *_split_halves(flow)])
                occs.append([*_split_halves(occ_cont), *_split_halves(occ)])
            else:
                flow = Fn.resize_bilinear_ac(flow, h, w)
                flows.append(list(_split_halves(flow)))
                xo_warp = warp_other(x, flow)
                flow_o_warp = warp_other(flow, flow)
                if l != self.num_levels - 1:
                    # conv_1x1_1 on x and on the warped other image: one launch over the 4B batch
                    both = self.conv_1x1_1(torch.cat([x, xo_warp], dim=0))
                    x_in, xo_w_in = _split_halves(both)
                else:
                    x_in, xo_w_in = x, xo_warp
                occ = self.occ_shuffle_upsample(occ, [x_in, xo_w_in, flow, flow_o_warp])
                occs.append(list(_split_halves(occ)))
            if l == self.__dict__.get("_debug_last_level", -1):      # tools/level_times.py: truncated passes (never set otherwise)
                break

        if self.training:
            return {'flow': flows, 'occ': occs}
        out_flow = Fn.resize_bilinear_ac(flow[:B], H, W, alpha=1.0 / div)
        out_occ = Fn.resize_bilinear_ac(occ[:B], H, W)
        return {'flow': out_flow, 'occ': out_occ}
