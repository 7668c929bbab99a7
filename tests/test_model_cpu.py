"""CPU-only host-logic checks: module surface, state_dict keys and MSRA-init parity with the reference."""
import os
import types

import numpy as np
import pytest
import torch

import irr_amd
from oracle import irr_pwc_oracle as O


def _args(bs=2):
    return types.SimpleNamespace(batch_size=bs, model_div_flow=0.05)


def test_state_dict_keys_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "init_seed0.npz"))
    torch.manual_seed(0)
    m = irr_amd.IRR_PWC(_args())
    sd = m.state_dict()
    assert list(sd.keys()) == [str(n) for n in g["names"]]
    assert [str(tuple(v.shape)) for v in sd.values()] == [str(s) for s in g["shapes"]]
    assert sum(v.numel() for v in sd.values()) == int(g["n_params"][0]) == 6362092
    # same RNG consumption order => bit-identical MSRA init under the same seed
    np.testing.assert_allclose([float(v.double().sum()) for v in sd.values()], g["sums"], rtol=0, atol=0)
    np.testing.assert_allclose([float(v.double().abs().sum()) for v in sd.values()], g["abssums"], rtol=0, atol=0)


def test_attributes_and_oracle_inventory():
    m = irr_amd.PWCNet(_args())
    assert m._div_flow == 0.05 and m.search_range == 4 and m.output_level == 4 and m.num_levels == 7
    assert m.num_chs == [3, 16, 32, 64, 96, 128, 196]
    assert m.corr_params == {"pad_size": 4, "kernel_size": 1, "max_disp": 4, "stride1": 1, "stride2": 1, "corr_multiply": 1}
    P = O.synthetic_params(0)
    assert set(P.keys()) == set(m.state_dict().keys())
    m.load_state_dict(P, strict=True)


def test_no_cpu_fallback():
    m = irr_amd.PWCNet(_args())
    x = torch.rand(1, 3, 64, 64)
    with pytest.raises(RuntimeError):
        m({"input1": x, "input2": x})


def test_correlation_signature():
    c = irr_amd.Correlation(pad_size=4, kernel_size=1, max_displacement=4, stride1=1, stride2=1, corr_multiply=1)
    assert (c.pad_size, c.kernel_size, c.max_displacement, c.stride1, c.stride2, c.corr_multiply) == (4, 1, 4, 1, 1, 1)
    # round 5: every parameter point of the legacy operator is accepted (FlowNet's (3, 3, 20, 1, 2) among them) ...
    c = irr_amd.Correlation(pad_size=3, kernel_size=3, max_displacement=20, stride1=1, stride2=2)
    assert (c.pad_size, c.kernel_size, c.max_displacement, c.stride1, c.stride2) == (3, 3, 20, 1, 2)
    # ... except what the reference's kernels cannot compute either: its DEFAULT kernel_size 0 divides by zero, other corr types do not exist
    for bad in (dict(), dict(kernel_size=2, max_displacement=1), dict(kernel_size=1, corr_multiply=0), dict(kernel_size=1, stride2=0)):
        with pytest.raises(ValueError):
            irr_amd.Correlation(**bad)
    with pytest.raises(RuntimeError):                       # no CPU fallback on this path either
        c(torch.rand(1, 2, 8, 8), torch.rand(1, 2, 8, 8))


def test_loss_scalar_algebra_matches_oracle():
    """losses.py:560-571: the balancing of the two terms (host logic; the per-pixel parts are HIP kernels and are
    checked on the GPU in tests/test_e2e_gpu.py)."""
    from irr_amd.losses import balance_and_total
    for f, o in ((3.0, 7.0), (9.0, 2.0), (4.0, 4.0)):
        fl, ol = torch.tensor(f, requires_grad=True), torch.tensor(o, requires_grad=True)
        got = balance_and_total(fl, ol, 4)
        if f > o:
            want = (f * 1 + o * (f / o)) / 4
        else:
            want = (f * (o / f) + o * 1) / 4
        np.testing.assert_allclose(float(got["total_loss"].detach()), want, rtol=1e-6)
        np.testing.assert_allclose(float(got["flow_loss"].detach()), f / 4, rtol=1e-6)
        got["total_loss"].backward()
        # the weights are detached: d total / d flow_loss = w_f / batch
        np.testing.assert_allclose(float(fl.grad), (1.0 if f > o else o / f) / 4, rtol=1e-6)


def test_loss_rejects_cpu_tensors():
    mod = irr_amd.MultiScaleEPE_PWC_Bi_Occ_upsample(_args(1)).train()
    out = {"flow": [[torch.zeros(1, 2, 8, 8), torch.zeros(1, 2, 8, 8)]], "occ": [[torch.zeros(1, 1, 8, 8), torch.zeros(1, 1, 8, 8)]]}
    tgt = {"target1": torch.zeros(1, 2, 8, 8), "target2": torch.zeros(1, 2, 8, 8),
           "target_occ1": torch.zeros(1, 1, 8, 8), "target_occ2": torch.zeros(1, 1, 8, 8)}
    with pytest.raises(RuntimeError):
        mod(out, tgt)
