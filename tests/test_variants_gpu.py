"""the seven pwcnet_* ablation models and the remaining PWC loss classes against vectors produced
by the imported reference (tests/golden/variants.npz, oracle/gen_golden.py::gen_variants; reference MSRA init under
seed 0, robust-mask protocol of SURVEY 8(c))."""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

VARIANTS = [("PWCNet_bi", "MultiScaleEPE_PWC_Bi"), ("PWCNet_occ", "MultiScaleEPE_PWC_Occ"), ("PWCNet_occ_bi", "MultiScaleEPE_PWC_Bi_Occ"),
            ("PWCNet_irr", "MultiScaleEPE_PWC"), ("PWCNet_irr_bi", "MultiScaleEPE_PWC_Bi"), ("PWCNet_irr_occ", "MultiScaleEPE_PWC_Occ"),
            ("PWCNet_irr_occ_bi", "MultiScaleEPE_PWC_Bi_Occ")]


def _batch(B=1, H=128, W=192, seed=4321):
    g = torch.Generator().manual_seed(seed)
    b = {"input1": torch.rand(B, 3, H, W, generator=g), "input2": torch.rand(B, 3, H, W, generator=g),
         "target1": 5 * torch.randn(B, 2, H, W, generator=g), "target2": 5 * torch.randn(B, 2, H, W, generator=g),
         "target_occ1": (torch.rand(B, 1, H, W, generator=g) < 0.2).float(),
         "target_occ2": (torch.rand(B, 1, H, W, generator=g) < 0.2).float(),
         "input_valid": (torch.rand(B, 1, H, W, generator=g) < 0.6).float()}
    return {k: v.cuda() for k, v in b.items()}


def _args():
    return types.SimpleNamespace(batch_size=1, model_div_flow=0.05)


@pytest.mark.parametrize("mname,lname", VARIANTS, ids=[v[0] for v in VARIANTS])
def test_ablation_model_matches_reference(golden_dir, mname, lname):
    import irr_amd
    g = np.load(os.path.join(golden_dir, "variants.npz"))
    batch = _batch()
    torch.manual_seed(0)
    m = getattr(irr_amd, mname)(_args(), mask_threshold=0.9999).cuda().eval()
    with torch.no_grad():
        out = m(batch)
    ref = torch.from_numpy(g[f"{mname}_flow"]).cuda()
    epe = torch.norm(out["flow"][:, :, ::2, ::2] - ref, dim=1).mean().item()
    scale = float(g[f"{mname}_flow_stats"][1])
    print(f"{mname}: eval EPE vs reference {epe:.3e} px (mean |flow| {scale:.2f} px)")
    assert out["flow"].shape == (1, 2, 128, 192)
    assert epe <= 1e-4 * max(1.0, scale / 5.0), epe            # MSRA-initialised nets output tens of pixels
    if f"{mname}_occ" in g.files:
        d = (out["occ"][:, :, ::2, ::2] - torch.from_numpy(g[f"{mname}_occ"]).cuda()).abs().mean().item()
        assert d <= 1e-4 * max(1.0, float(np.abs(g[f"{mname}_occ"]).mean())), d
    m.train()
    loss = getattr(irr_amd, lname)(_args()).train()
    ld = loss(m(batch), batch)
    ld["total_loss"].backward()
    gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters() if p.grad is not None)))
    want = g[f"{mname}_losses"]
    got = [float(ld.get("flow_loss", ld["total_loss"]).detach()), float(ld["occ_loss"].detach()) if "occ_loss" in ld else 0.0,
           float(ld["total_loss"].detach()), gn]
    np.testing.assert_allclose(got[:3], want[:3], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(got[3], want[3], rtol=2e-3)


@pytest.mark.parametrize("lname", ["MultiScaleEPE_PWC_Bi_Occ_upsample_Sintel", "MultiScaleEPE_PWC_Bi_Occ_upsample_KITTI"])
def test_finetuning_losses_match_reference(golden_dir, lname):
    import irr_amd
    g = np.load(os.path.join(golden_dir, "variants.npz"))
    batch = _batch()
    torch.manual_seed(0)
    m = irr_amd.PWCNet(_args(), mask_threshold=0.9999).cuda().train()
    ld = getattr(irr_amd, lname)(_args()).train()(m(batch), batch)
    ld["total_loss"].backward()
    gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters() if p.grad is not None)))
    want = g[f"{lname}_losses"]
    got = [float(ld["flow_loss"].detach()), float(ld["occ_loss"].detach()) if "occ_loss" in ld else 0.0, float(ld["total_loss"].detach()), gn]
    np.testing.assert_allclose(got[:3], want[:3], rtol=5e-5, atol=1e-6)
    np.testing.assert_allclose(got[3], want[3], rtol=2e-3)


def test_kitti_eval_metrics(golden_dir):
    import irr_amd
    g = np.load(os.path.join(golden_dir, "variants.npz"))
    batch = _batch()
    pred = {"flow": torch.from_numpy(g["kitti_eval_pred_flow"]).cuda(), "occ": torch.zeros(1, 1, 128, 192).cuda()}
    le = irr_amd.MultiScaleEPE_PWC_Bi_Occ_upsample_KITTI(_args()).eval()(pred, batch)
    np.testing.assert_allclose([float(le["epe"]), float(le["outlier"])], g["kitti_eval"], rtol=1e-5)


def test_async_wgrad_lane_is_race_free_on_irr_variant():
    """PWCNet_irr_occ_bi adds residuals OUTSIDE the decoder nodes (``flow = flow + flow_res``), so one gradient tensor
    reaches the dense-estimator node (whose conv_last weight gradient the asynchronous lane reads) AND the previous level's
    producer, into which the autograd engine accumulates in place when it owns the last reference.  The lane keeps its
    operands alive until it has passed them; every gradient of repeated two-stream backward passes must equal the
    single-stream gradients of the same inputs."""
    import irr_amd
    from irr_amd import ddp
    torch.manual_seed(0)
    m = irr_amd.PWCNet_irr_occ_bi(_args(), mask_threshold=0.9999).cuda().train()
    loss = irr_amd.MultiScaleEPE_PWC_Bi_Occ(_args()).train()
    arena = ddp.GradArena(m.named_parameters())
    b = _batch(B=2)

    def grads(lane):
        if lane:
            arena.enable_async_wgrad()
        try:
            arena.zero_grad()
            loss(m(b), b)["total_loss"].backward()
            arena.sync()
            torch.cuda.synchronize()
            return {n: p.grad.detach().clone() for n, p in m.named_parameters()}
        finally:
            if lane:
                arena.disable_async_wgrad()

    ref = grads(False)
    for it in range(8):
        g = grads(True)
        for n, r in ref.items():
            d = (g[n] - r).double().norm().item()
            assert d <= 1e-3 * r.double().norm().item() + 1e-7, (it, n, d)
