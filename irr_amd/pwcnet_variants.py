"""The PWC-Net ablation ladder of the reference (models/pwcnet_bi.py, pwcnet_occ.py, pwcnet_occ_bi.py, pwcnet_irr.py,
pwcnet_irr_bi.py, pwcnet_irr_occ.py, pwcnet_irr_occ_bi.py; registry names models/__init__.py:27-34) on the same
MI355X kernels as IRR-PWC.  One configurable implementation, seven classes with the reference's constructor
(``args, div_flow=0.05``), forward contract (``input_dict -> {'flow': ..., ['occ': ...]}``; lists of per-level outputs
in training mode, full-resolution tensors in eval mode) and ``state_dict`` keys.

Three switches span the ladder:
  irr -- ONE shared flow (and occlusion) decoder + context network applied at every level on 32-channel 1x1-projected
         features, flow carried in level-local units between levels (pwcnet_irr.py:62-75) -- versus per-level decoders on
         the raw pyramid features and a context network at the output level only (pwcnet.py:25-38);
  occ -- an occlusion decoder / context network next to the flow ones (pwcnet_occ.py:23-43);
  bi  -- both directions (x1->x2 and x2->x1) with shared weights (pwcnet_bi.py:62-92).

``rescale_flow`` of the reference multiplies its argument in place and returns a copy (models/pwc_modules.py:70-82);
in all seven files the result is bound back to the same name, so the mutation is unobservable and the pure
``modules.rescale_flow`` is an exact restatement (unlike models/IRR_PWC.py).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import functional as Fn
from .modules import (ContextNetwork, FeatureExtractor, FlowEstimatorDense, OccContextNetwork, OccEstimatorDense,
                      WarpingLayer, conv, initialize_msra, rescale_flow)


class _PWCVariant(nn.Module):
    IRR = False
    OCC = False
    BI = False

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
        if self.IRR:
            # pwcnet_irr.py:23-36 / pwcnet_irr_occ.py:23-41
            num_ch_in_flo = self.dim_corr + 32 + 2
            self.flow_estimators = FlowEstimatorDense(num_ch_in_flo)
            self.context_networks = ContextNetwork(num_ch_in_flo + 448 + 2)
            if self.OCC:
                num_ch_in_occ = self.dim_corr + 32 + 1
                self.occ_estimators = OccEstimatorDense(num_ch_in_occ)
                self.occ_context_networks = OccContextNetwork(num_ch_in_occ + 448 + 1)
            self.conv_1x1 = nn.ModuleList([conv(c, 32, kernel_size=1, stride=1, dilation=1) for c in (196, 128, 96, 64, 32)])
        else:
            # pwcnet_bi.py:23-37 / pwcnet_occ.py:23-43: one decoder per level, context network(s) for the output level
            self.flow_estimators = nn.ModuleList()
            if self.OCC:
                self.occ_estimators = nn.ModuleList()
            for l, ch in enumerate(self.num_chs[::-1]):
                if l > self.output_level:
                    break
                self.flow_estimators.append(FlowEstimatorDense(self.dim_corr if l == 0 else self.dim_corr + ch + 2))
                if self.OCC:
                    self.occ_estimators.append(OccEstimatorDense(self.dim_corr if l == 0 else self.dim_corr + ch + 1))
            self.context_networks = ContextNetwork(self.dim_corr + 32 + 2 + 448 + 2)
            if self.OCC:
                self.context_networks_occ = OccContextNetwork(self.dim_corr + 32 + 1 + 448 + 1)
        self.corr_params = {"pad_size": self.search_range, "kernel_size": 1, "max_disp": self.search_range,
                            "stride1": 1, "stride2": 1, "corr_multiply": 1}
        initialize_msra(self.modules())

    def forward(self, input_dict):
        x1_raw, x2_raw = input_dict['input1'], input_dict['input2']
        B, _, H, W = x1_raw.shape
        both = self.feature_pyramid_extractor(torch.cat([x1_raw, x2_raw], dim=0))     # both images in one pass
        ndir = 2 if self.BI else 1
        flow = [None] * ndir
        occ = [None] * ndir
        flows, occs = [], []
        for l, feat in enumerate(both):
            xs = (feat[:B], feat[B:])
            h, w = feat.shape[2], feat.shape[3]
            last = l == self.output_level
            for d in range(ndir):
                xa, xb = xs[d], xs[1 - d]                  # direction 0: x1 -> x2, direction 1: x2 -> x1
                if l == 0:
                    xb_warp = xb
                    flow[d] = torch.zeros(B, 2, h, w, device=feat.device, dtype=torch.float32)
                    occ[d] = torch.zeros(B, 1, h, w, device=feat.device, dtype=torch.float32)
                else:
                    flow[d] = Fn.resize_bilinear_ac(flow[d], h, w)
                    if self.OCC:
                        occ[d] = Fn.resize_bilinear_ac(occ[d], h, w)
                    xb_warp = self.warping_layer(xb, flow[d], H, W, self._div_flow)
                corr = Fn.cost_volume(xa, xb_warp, lrelu=True)
                if self.IRR:
                    # shared decoders on projected features; flow in level-local units inside the level
                    f_loc = rescale_flow(flow[d], self._div_flow, W, H, to_local=True)
                    xa_1by1 = self.conv_1x1[l](xa)
                    x_intm, f_res = self.flow_estimators(torch.cat([corr, xa_1by1, f_loc], dim=1))
                    f_loc = f_loc + f_res
                    f_loc = self.context_networks(torch.cat([x_intm, f_loc], dim=1), res=f_loc)
                    flow[d] = rescale_flow(f_loc, self._div_flow, W, H, to_local=False)
                    if self.OCC:
                        x_intm_o, o_res = self.occ_estimators(torch.cat([corr, xa_1by1, occ[d]], dim=1))
                        o = occ[d] + o_res
                        occ[d] = self.occ_context_networks(torch.cat([x_intm_o, o], dim=1), res=o)
                else:
                    if l == 0:
                        x_intm, flow[d] = self.flow_estimators[l](corr)
                        if self.OCC:
                            x_intm_o, occ[d] = self.occ_estimators[l](corr)
                    else:
                        x_intm, flow[d] = self.flow_estimators[l](torch.cat([corr, xa, flow[d]], dim=1))
                        if self.OCC:
                            # pwcnet_occ_bi.py:101: the backward occlusion decoder is fed x1, not x2 (kept as is)
                            xo = xs[0] if (self.BI and d == 1) else xa
                            x_intm_o, occ[d] = self.occ_estimators[l](torch.cat([corr, xo, occ[d]], dim=1))
                    if last:
                        flow[d] = self.context_networks(torch.cat([x_intm, flow[d]], dim=1), res=flow[d])
                        if self.OCC:
                            occ[d] = self.context_networks_occ(torch.cat([x_intm_o, occ[d]], dim=1), res=occ[d])
            flows.append([flow[0], flow[1]] if self.BI else flow[0])
            if self.OCC:
                occs.append([occ[0], occ[1]] if self.BI else occ[0])
            if last:
                break
        if self.training:
            out = {'flow': flows}
            if self.OCC:
                out['occ'] = occs
            return out
        out = {'flow': Fn.resize_bilinear_ac(flow[0], H, W, alpha=1.0 / self._div_flow)}
        if self.OCC:
            out['occ'] = Fn.resize_bilinear_ac(occ[0], H, W)
        return out


class PWCNet_bi(_PWCVariant):
    """models/pwcnet_bi.py"""
    BI = True


class PWCNet_occ(_PWCVariant):
    """models/pwcnet_occ.py"""
    OCC = True


class PWCNet_occ_bi(_PWCVariant):
    """models/pwcnet_occ_bi.py"""
    OCC = True
    BI = True


class PWCNet_irr(_PWCVariant):
    """models/pwcnet_irr.py"""
    IRR = True


class PWCNet_irr_bi(_PWCVariant):
    """models/pwcnet_irr_bi.py"""
    IRR = True
    BI = True


class PWCNet_irr_occ(_PWCVariant):
    """models/pwcnet_irr_occ.py"""
    IRR = True
    OCC = True


class PWCNet_irr_occ_bi(_PWCVariant):
    """models/pwcnet_irr_occ_bi.py"""
    IRR = True
    OCC = True
    BI = True
