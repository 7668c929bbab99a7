"""PWC-Net baseline (models/pwcnet.py of the reference) on the same MI355X kernels -- BASELINE config 0
("plumbing"): per-level dense flow estimators, one context network at the output level, single direction.
Same constructor / forward contract / state_dict keys as the reference's ``models.PWCNet``."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import functional as Fn
from .modules import ContextNetwork, FeatureExtractor, FlowEstimatorDense, WarpingLayer, initialize_msra


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
        self.flow_estimators = nn.ModuleList()
        self.dim_corr = (self.search_range * 2 + 1) ** 2
        for l, ch in enumerate(self.num_chs[::-1]):                  # models/pwcnet.py:25-35
            if l > self.output_level:
                break
            num_ch_in = self.dim_corr if l == 0 else self.dim_corr + ch + 2
            self.flow_estimators.append(FlowEstimatorDense(num_ch_in))
        self.context_networks = ContextNetwork(self.dim_corr + 32 + 2 + 448 + 2)
        self.corr_params = {"pad_size": self.search_range, "kernel_size": 1, "max_disp": self.search_range,
                            "stride1": 1, "stride2": 1, "corr_multiply": 1}
        initialize_msra(self.modules())

    def forward(self, input_dict):
        x1_raw, x2_raw = input_dict['input1'], input_dict['input2']
        B, _, H, W = x1_raw.shape
        both = self.feature_pyramid_extractor(torch.cat([x1_raw, x2_raw], dim=0))     # both images in one pass
        flows = []
        flow = None
        for l, feat in enumerate(both):
            x1, x2 = feat[:B], feat[B:]
            if l == 0:
                x2_warp = x2
            else:
                flow = Fn.resize_bilinear_ac(flow, x1.shape[2], x1.shape[3])
                x2_warp = self.warping_layer(x2, flow, H, W, self._div_flow)
            corr = Fn.cost_volume(x1, x2_warp, lrelu=True)
            inp = corr if l == 0 else torch.cat([corr, x1, flow], dim=1)
            x_intm, flow = self.flow_estimators[l](inp)
            if l != self.output_level:
                flows.append(flow)
            else:
                flow = self.context_networks(torch.cat([x_intm, flow], dim=1), res=flow)
                flows.append(flow)
                break
        if self.training:
            return {'flow': flows}
        return {'flow': Fn.resize_bilinear_ac(flow, H, W, alpha=1.0 / self._div_flow)}
