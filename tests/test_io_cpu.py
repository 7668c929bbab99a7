"""CPU: .flo round trip (byte layout of utils/flow.py:11-34) and reference-compatible checkpoints."""
import os
import types

import numpy as np
import pytest
import torch

import irr_amd
from irr_amd import io as I
from irr_amd.train import ModelAndLoss


def test_flo_roundtrip_and_layout(tmp_path):
    rng = np.random.default_rng(0)
    uv = rng.standard_normal((5, 7, 2)).astype(np.float32)
    p = str(tmp_path / "a.flo")
    I.write_flow(p, uv)
    raw = open(p, "rb").read()
    assert len(raw) == 4 + 4 + 4 + 5 * 7 * 2 * 4
    assert np.frombuffer(raw[:4], np.float32)[0] == np.float32(202021.25)
    assert tuple(np.frombuffer(raw[4:12], np.int32)) == (7, 5)               # width first, then height
    np.testing.assert_array_equal(np.frombuffer(raw[12:], np.float32).reshape(5, 7, 2), uv)
    np.testing.assert_array_equal(I.read_flo_as_float32(p), uv)
    I.write_flow(p, uv[:, :, 0], uv[:, :, 1])
    np.testing.assert_array_equal(I.read_flo_as_float32(p), uv)
    I.flow_tensor_to_flo(p, torch.from_numpy(uv.transpose(2, 0, 1))[None])
    np.testing.assert_array_equal(I.read_flo_as_float32(p), uv)
    open(p, "wb").write(b"\\0" * 12)
    with pytest.raises(ValueError):
        I.read_flo_as_float32(p)


def test_checkpoint_reference_layout(tmp_path):
    args = types.SimpleNamespace(batch_size=2, model_div_flow=0.05)
    torch.manual_seed(1)
    m = irr_amd.PWCNet(args)
    mal = ModelAndLoss(args, m, irr_amd.MultiScaleEPE_PWC_Bi_Occ_upsample(args))
    saver = I.CheckpointSaver()
    f = saver.save_latest(str(tmp_path), mal, {"epoch": 3, "epe": 1.5}, store_as_best=True)
    assert os.path.basename(f) == "checkpoint_latest.ckpt" and os.path.exists(str(tmp_path / "checkpoint_best.ckpt"))
    ck = torch.load(f, weights_only=False)
    assert set(ck.keys()) == {"epoch", "epe", "state_dict"}
    assert len(ck["state_dict"]) == 124 and all(k.startswith("_model.") for k in ck["state_dict"])
    # restore into a differently initialised model, excluding the refinement heads
    torch.manual_seed(2)
    m2 = irr_amd.PWCNet(args)
    mal2 = ModelAndLoss(args, m2, irr_amd.MultiScaleEPE_PWC_Bi_Occ_upsample(args))
    before = m2.refine_flow.convs[0][0].weight.clone()
    stats, _ = saver.restore_latest(str(tmp_path), mal2, include_params="*", exclude_params=["_model.refine_*"])
    assert stats == {"epoch": 3, "epe": 1.5}
    assert torch.equal(m2.flow_estimators.conv1[0].weight, m.flow_estimators.conv1[0].weight)
    assert torch.equal(m2.refine_flow.convs[0][0].weight, before)
