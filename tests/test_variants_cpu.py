"""Host-side surface of the ablation ladder: class names, state_dict keys and the MSRA
initialisation under seed 0 equal the reference's (recorded by oracle/gen_golden.py::gen_variants)."""
import os
import types

import numpy as np
import pytest
import torch

NAMES = ["PWCNet_bi", "PWCNet_occ", "PWCNet_occ_bi", "PWCNet_irr", "PWCNet_irr_bi", "PWCNet_irr_occ", "PWCNet_irr_occ_bi"]


@pytest.mark.parametrize("name", NAMES)
def test_variant_surface_and_init(golden_dir, name):
    import irr_amd
    g = np.load(os.path.join(golden_dir, "variants.npz"))
    args = types.SimpleNamespace(batch_size=1, model_div_flow=0.05)
    torch.manual_seed(0)
    m = getattr(irr_amd, name)(args)
    sd = m.state_dict()
    assert list(sd.keys()) == [str(k) for k in g[f"{name}_keys"]]
    got = [float(sum(v.double().sum() for v in sd.values())), float(sum(v.double().abs().sum() for v in sd.values()))]
    np.testing.assert_allclose(got, g[f"{name}_init"], rtol=1e-12)
    assert m._div_flow == 0.05 and m.search_range == 4 and m.output_level == 4 and m.corr_params["max_disp"] == 4
    with pytest.raises(RuntimeError):                       # no CPU fallback on the product path
        m.eval()({"input1": torch.rand(1, 3, 64, 64), "input2": torch.rand(1, 3, 64, 64)})


def test_loss_classes_exported():
    import irr_amd
    args = types.SimpleNamespace(batch_size=2, model_div_flow=0.05)
    for n in ("MultiScaleEPE_PWC", "MultiScaleEPE_PWC_Bi", "MultiScaleEPE_PWC_Occ", "MultiScaleEPE_PWC_Bi_Occ",
              "MultiScaleEPE_PWC_Bi_Occ_upsample", "MultiScaleEPE_PWC_Bi_Occ_upsample_Sintel", "MultiScaleEPE_PWC_Bi_Occ_upsample_KITTI"):
        assert getattr(irr_amd, n)(args)._batch_size == 2
