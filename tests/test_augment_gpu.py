"""GPU parity of irr_amd.augment.RandomAffineFlowOcc (kernels irr_affine_warp_f32 / irr_affine_flow_occ_f32) against the
golden vectors of the imported reference (tests/golden/augment.npz) and against oracle/augment_oracle.py on seeded inputs."""
import os
import sys
import types

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import augment_oracle as AO  # noqa: E402
from test_augment_cpu import KEYS, load_case  # noqa: E402

pytestmark = pytest.mark.gpu

# coordinates are reproduced with the reference's fp32 op order (-ffp-contract=off); what remains is the last-ulp
# difference of sin/cos/division between host libraries -- none in practice.  Flow values are O(10) px.
TOL = {"input1": 2e-6, "input2": 2e-6, "target1": 2e-4, "target2": 2e-4, "target_occ1": 2e-6, "target_occ2": 2e-6}


def _run(ex, noise, crop, seed):
    from irr_amd.augment import RandomAffineFlowOcc
    aug = RandomAffineFlowOcc(types.SimpleNamespace(), addnoise=noise, crop=crop)
    torch.manual_seed(seed)
    np.random.seed(seed)
    return aug({k: v.cuda() for k, v in ex.items()})


@pytest.mark.parametrize("name", ["plain", "crop"])
def test_matches_reference_golden(name):
    ex, out, th, noise, crop, seed = load_case(name)
    got = _run(ex, noise, crop, seed)
    for k in KEYS:
        g = got[k].cpu()
        assert g.shape == out[k].shape, k
        err = (g - out[k]).abs().max().item()
        assert err <= TOL[k], (k, err)
    # the occlusion maps are {0,1}-blends: bit-exact agreement of the hard 0/1 pixels
    assert torch.equal(got["target_occ1"].cpu() == 1.0, out["target_occ1"] == 1.0)


def test_noise_statistics_and_clamp():
    ex, out, th, noise, crop, seed = load_case("noise")
    from irr_amd.augment import RandomAffineFlowOcc
    B, _, H, W = ex["input1"].shape
    torch.manual_seed(seed)
    np.random.seed(seed)
    clean = RandomAffineFlowOcc(types.SimpleNamespace(), addnoise=False)({k: v.cuda() for k, v in ex.items()})
    torch.manual_seed(seed)
    np.random.seed(seed)
    noisy = RandomAffineFlowOcc(types.SimpleNamespace(), addnoise=True)({k: v.cuda() for k, v in ex.items()})
    np.random.seed(seed)
    std = np.random.uniform(0.0, 0.04)
    for k in ("target1", "target2", "target_occ1", "target_occ2"):       # geometry is untouched by the noise
        assert torch.equal(clean[k], noisy[k]), k
        assert (noisy[k].cpu() - out[k]).abs().max().item() <= TOL[k]
    for k in ("input1", "input2"):
        assert noisy[k].min().item() >= 0.0 and noisy[k].max().item() <= 1.0
        interior = (clean[k] > 0.2) & (clean[k] < 0.8)                    # away from the clamp
        d = (noisy[k] - clean[k])[interior]
        assert abs(d.mean().item()) < 4 * std / d.numel() ** 0.5 + 1e-6
        assert abs(d.std().item() / std - 1.0) < 0.1


@pytest.mark.parametrize("B,H,W,crop", [(4, 96, 128, None), (2, 128, 192, [64, 128])])
def test_matches_oracle_seeded(B, H, W, crop):
    g = torch.Generator().manual_seed(7)
    ex = {"input1": torch.rand(B, 3, H, W, generator=g), "input2": torch.rand(B, 3, H, W, generator=g),
          "target1": 6 * torch.randn(B, 2, H, W, generator=g), "target2": 6 * torch.randn(B, 2, H, W, generator=g),
          "target_occ1": (torch.rand(B, 1, H, W, generator=g) < 0.3).float(),
          "target_occ2": (torch.rand(B, 1, H, W, generator=g) < 0.3).float()}
    torch.manual_seed(3)
    np.random.seed(3)
    want, _ = AO.random_affine_flow_occ({k: v.clone() for k, v in ex.items()}, addnoise=False, crop=crop)
    got = _run(ex, False, crop, 3)
    for k in KEYS:
        err = (got[k].cpu() - want[k]).abs().max().item()
        assert err <= TOL[k], (k, err)


def test_properties_full_size():
    """BASELINE-size batch (8 x 384 x 448): identity thetas reproduce the input; a pure mirror is an exact flip."""
    from irr_amd.augment import RandomAffineFlowOcc
    B, H, W = 8, 384, 448
    g = torch.Generator().manual_seed(1)
    ex = {"input1": torch.rand(B, 3, H, W, generator=g).cuda(), "input2": torch.rand(B, 3, H, W, generator=g).cuda(),
          "target1": torch.zeros(B, 2, H, W).cuda(), "target2": torch.zeros(B, 2, H, W).cuda(),
          "target_occ1": torch.zeros(B, 1, H, W).cuda(), "target_occ2": torch.zeros(B, 1, H, W).cuda()}
    aug = RandomAffineFlowOcc(types.SimpleNamespace(), addnoise=False)
    ident = torch.tensor([[1.0, 0, 0, 0, 1.0, 0]]).repeat(B, 1)
    out = aug(dict(ex), thetas=(ident, ident))
    assert (out["input1"] - ex["input1"]).abs().max().item() <= 1e-4      # coordinates round-trip to ~1e-5 px
    assert out["target1"].abs().max().item() <= 1e-3
    flip = ident * torch.tensor([[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0]])
    out = aug(dict(ex), thetas=(flip, flip))
    # (the outermost columns may fall an ulp outside the frame and read as zero, exactly as in the reference)
    assert (out["input2"] - ex["input2"].flip(3))[..., 1:-1].abs().max().item() <= 1e-4


def test_cpu_tensor_raises():
    from irr_amd import hip
    from irr_amd.augment import affine_warp
    with pytest.raises(hip.HipError):
        affine_warp(torch.zeros(1, 3, 8, 8), torch.zeros(1, 6))
