"""Pins oracle/augment_oracle.py (CPU restatement of RandomAffineFlowOcc, augmentations.py:368-653) against
tests/golden/augment.npz, which oracle/gen_golden.py produced from the imported reference with the same seeds."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import augment_oracle as AO  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden", "augment.npz")
KEYS = ["input1", "input2", "target1", "target2", "target_occ1", "target_occ2"]


def load_case(name):
    g = np.load(GOLD)
    B, H, W, noise, ch, cw, seed = [int(v) for v in g[f"{name}_cfg"]]
    ex = {k: torch.from_numpy(g[f"{name}_in_{k}"]) for k in KEYS}
    out = {k: torch.from_numpy(g[f"{name}_out_{k}"]) for k in KEYS}
    th = (torch.from_numpy(g[f"{name}_theta1_sampled"]), torch.from_numpy(g[f"{name}_theta2_sampled"]))
    return ex, out, th, bool(noise), ([ch, cw] if ch else None), seed


@pytest.mark.parametrize("name", ["plain", "crop", "noise"])
def test_oracle_matches_reference(name):
    ex, out, th, noise, crop, seed = load_case(name)
    torch.manual_seed(seed)
    np.random.seed(seed)
    got, _ = AO.random_affine_flow_occ(ex, addnoise=noise, crop=crop)
    for k in KEYS:
        assert got[k].shape == out[k].shape, k
        # same RNG stream, same arithmetic -> identical up to libm/FMA differences of the torch build (none seen)
        err = (got[k] - out[k]).abs().max().item()
        assert err <= 1e-5, (k, err)


def test_sampled_thetas_reproduced():
    ex, out, th, noise, crop, seed = load_case("plain")
    torch.manual_seed(seed)
    B, _, H, W = ex["input1"].shape
    theta0 = torch.tensor([[1.0, 0.0, 0.0, 0.0, 1.0, 0.0]]).repeat(B, 1)
    t1 = AO.sample_thetas(theta0, 0.2, 1.0, 1.5, 0.86, 1.16, -0.2, 0.2, [H, W])
    t2 = AO.sample_thetas(t1, 0.015, 0.985, 1.015, 1.0, 1.0, -0.015, 0.015, [H, W])
    assert torch.equal(t1, th[0]) and torch.equal(t2, th[1])
    assert not AO.find_invalid(W, H, t1).any()
