"""Inputs whose height / width are not multiples of 64 (Sintel 436x1024, KITTI ~375x1242: the reference evaluates them as they
come, scripts/validation/IRR-PWC_sintel.sh:17-29): odd pyramid sizes, the align_corners=False fallback of upsample_factor2
(models/irr_modules.py:21-27), adaptive pooling with non-integer ratios in the loss -- against vectors produced by the
imported reference (tests/golden/oddsize.npz, oracle/gen_golden.py::gen_oddsize; robust-mask protocol)."""
import os
import types

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _args(bs=2):
    return types.SimpleNamespace(batch_size=bs, model_div_flow=0.05)


def _model(train=False):
    import irr_amd
    from oracle import irr_pwc_oracle as O
    m = irr_amd.PWCNet(_args(), mask_threshold=0.9999)
    m.load_state_dict(O.synthetic_params(0), strict=True)
    m = m.cuda()
    return m.train() if train else m.eval()


def _batch(B, H, W):
    from oracle import irr_pwc_oracle as O
    return {k: v.cuda() for k, v in O.synthetic_batch(B, H, W, 1234).items()}


@pytest.mark.parametrize("H,W", [(436, 1024), (375, 1242)])
def test_eval_native_dataset_sizes_vs_reference(golden_dir, H, W):
    """Sintel- and KITTI-sized pairs, B = 1: 4096 sampled output pixels of the full-resolution flow / occlusion maps"""
    g = np.load(os.path.join(golden_dir, "oddsize.npz"))
    k = f"{H}x{W}"
    m = _model()
    b = _batch(1, H, W)
    with torch.no_grad():
        out = m({"input1": b["input1"], "input2": b["input2"]})
    assert out["flow"].shape == (1, 2, H, W) and out["occ"].shape == (1, 1, H, W)
    idx = torch.from_numpy(g[k + "_idx"]).cuda()
    epe = torch.norm(out["flow"].reshape(1, 2, -1)[:, :, idx] - torch.from_numpy(g[k + "_flow_samples"]).cuda(), dim=1).mean().item()
    oc = (out["occ"].reshape(1, 1, -1)[:, :, idx] - torch.from_numpy(g[k + "_occ_samples"]).cuda()).abs().mean().item()
    print(f"{k} robust-mask eval: EPE vs reference {epe:.3e} px, occ logit diff {oc:.3e}")
    assert epe <= 1e-4 and oc <= 1e-4, (epe, oc)


def test_odd_pyramid_eval_and_train_step_vs_reference(golden_dir, routing):
    """100x132 (levels 50x66, 25x33, 13x17, 7x9, 4x5, 2x3), B = 2: eval outputs in full, then one train step with the
    general adaptive pooling in the loss: losses, 124 gradient norms, post-Adam parameter checksums"""
    import irr_amd
    from irr_amd.train import ModelAndLoss, TrainStep, make_adam
    g = np.load(os.path.join(golden_dir, "oddsize.npz"))
    names = [str(n) for n in g["param_names"]]
    b = _batch(2, 100, 132)
    m = _model()
    with torch.no_grad():
        out = m({"input1": b["input1"], "input2": b["input2"]})
    epe = torch.norm(out["flow"] - torch.from_numpy(g["small_eval_flow"]).cuda(), dim=1).mean().item()
    oc = (out["occ"] - torch.from_numpy(g["small_eval_occ"]).cuda()).abs().mean().item()
    print(f"100x132 eval: EPE {epe:.3e} px, occ diff {oc:.3e}")
    assert epe <= 1e-4 and oc <= 1e-4, (epe, oc)
    m.train()
    o = m(b)
    assert [list(lv[0].shape[2:]) for lv in o["flow"]] == g["small_train_sizes"].tolist()
    loss = irr_amd.MultiScaleEPE_PWC_Bi_Occ_upsample(_args(2)).train()
    mal = ModelAndLoss(_args(2), m, loss).train()
    step = TrainStep(mal, make_adam(m.parameters()))
    ld, _, _ = step(b)
    got = np.array([float(ld["flow_loss"].detach()), float(ld["occ_loss"].detach()), float(ld["total_loss"].detach())])
    np.testing.assert_allclose(got, g["small_train_losses"], rtol=2e-5)
    sd = dict(m.named_parameters())
    gn = np.array([float(sd[n].grad.double().norm()) for n in names])
    ref = g["small_train_gradnorm"]
    tot_ref = np.sqrt((ref ** 2).sum())
    assert abs(np.sqrt((gn ** 2).sum()) - tot_ref) / tot_ref < 1e-4
    np.testing.assert_allclose(gn, ref, rtol=5e-3, atol=1e-4 * tot_ref)
    # Adam's first step moves every element by lr * sign(g): a gradient element that is zero up to rounding (many are, at
    # 2x3 ... 7x9 pixels) may take either sign, 2e-4 of checksum per flip -- allow 10 + 0.02 % of the elements to flip
    post = np.array([float(sd[n].detach().double().sum()) for n in names])
    flips = np.array([10 + 2e-4 * sd[n].numel() for n in names])
    assert (np.abs(post - g["small_poststep_sum"]) <= 1e-5 * np.abs(post) + 2e-4 * flips).all(), \
        np.abs(post - g["small_poststep_sum"]).max()


@pytest.mark.parametrize("shape,size", [((2, 3, 14, 18), (13, 17)), ((1, 1, 110, 132), (109, 131)), ((2, 2, 7, 9), (20, 31)), ((1, 2, 40, 40), (9, 5))])
def test_resize_bilinear_half_pixel_vs_torch(shape, size):
    """irr_resize_bilinear_hp_{fwd,bwd}: F.interpolate(..., mode='bilinear', align_corners=False) incl. its gradient"""
    from irr_amd import functional as Fn
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g)
    go = torch.randn(shape[0], shape[1], *size, generator=g)
    xc = x.clone().requires_grad_(True)
    yr = F.interpolate(xc, list(size), mode="bilinear", align_corners=False)
    yr.backward(go)
    xd = x.cuda().requires_grad_(True)
    y = Fn.resize_bilinear(xd, *size)
    y.backward(go.cuda())
    np.testing.assert_allclose(y.detach().cpu().numpy(), yr.detach().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(xd.grad.cpu().numpy(), xc.grad.numpy(), rtol=1e-5, atol=1e-5)


def test_upsample_factor2_fallback():
    from irr_amd import functional as Fn
    x = torch.randn(2, 1, 55, 64)
    for tgt in ((110, 128), (109, 128), (109, 127)):
        ref = F.interpolate(x, scale_factor=2, mode="nearest")
        if tuple(ref.shape[2:]) != tgt:
            ref = F.interpolate(ref, list(tgt), mode="bilinear", align_corners=False)
        out = Fn.upsample_factor2(x.cuda(), torch.empty(2, 3, *tgt))
        np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("hw,size", [((100, 132), (13, 17)), ((436, 1024), (109, 256)), ((375, 1242), (6, 20)), ((64, 128), (16, 32))])
def test_adaptive_avg_pool_vs_torch(hw, size):
    from irr_amd.losses import avg_pool_to
    x = torch.randn(2, 2, *hw, generator=torch.Generator().manual_seed(hw[0]))
    ref = 0.05 * F.adaptive_avg_pool2d(x, list(size))
    out = avg_pool_to(x.cuda(), size[0], size[1], 0.05)
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-7)
