"""Pins the oracle (oracle/irr_pwc_oracle.py) against vectors produced by the imported
reference (oracle/gen_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import irr_pwc_oracle as O

torch.set_num_threads(8)


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def T(a, grad=False):
    t = torch.from_numpy(np.array(a))
    return t.requires_grad_(True) if grad else t


@pytest.mark.parametrize("case", ["a", "b", "c", "d"])
def test_cost_volume_vs_reference(golden_dir, case):
    g = _load(golden_dir, "ops_basic.npz")
    f1, f2 = T(g[f"corr_{case}_f1"], True), T(g[f"corr_{case}_f2"], True)
    out = O.cost_volume(f1, f2)
    out.backward(T(g[f"corr_{case}_go"]))
    assert torch.equal(out.detach(), T(g[f"corr_{case}_out"]))
    np.testing.assert_allclose(f1.grad.numpy(), g[f"corr_{case}_g1"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(f2.grad.numpy(), g[f"corr_{case}_g2"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("case", ["a", "b", "c", "z"])
@pytest.mark.parametrize("mode,thr", [("asis", 1.0), ("robust", 0.9999)])
def test_warp_vs_reference(golden_dir, case, mode, thr):
    g = _load(golden_dir, "ops_basic.npz")
    k = f"warp_{case}_{mode}"
    H, W = [int(v) for v in g[k + "_HW"]]
    x, fl = T(g[k + "_x"], True), T(g[k + "_flow"], True)
    out = O.warp(x, fl, H, W, 0.05, thr)
    out.backward(T(g[k + "_go"]))
    assert torch.equal(out.detach(), T(g[k + "_out"]))
    np.testing.assert_allclose(x.grad.numpy(), g[k + "_gx"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(fl.grad.numpy(), g[k + "_gflow"], rtol=1e-6, atol=1e-7)
    with torch.no_grad():
        m = O.warp(torch.ones(x.shape[0], 1, *x.shape[2:]), fl.detach(), H, W, 0.05, thr)
    assert torch.equal(m, T(g[k + "_mask"]))


def test_modules_vs_reference(golden_dir):
    g = _load(golden_dir, "ops_modules.npz")
    P = O.synthetic_params(0)
    o = O.refine_flow(P, T(g["rf_flow"]), T(g["rf_dimg"]), T(g["rf_feat"]))
    np.testing.assert_allclose(o.numpy(), g["rf_out"], rtol=1e-6, atol=1e-6)
    o = O.refine_occ(P, T(g["ro_occ"]), T(g["ro_f1"]), T(g["ro_f2"]))
    np.testing.assert_allclose(o.numpy(), g["ro_out"], rtol=1e-6, atol=1e-6)
    o = O.occ_upsample(P, T(g["ou_occ"]), T(g["ou_guide"]))
    np.testing.assert_allclose(o.numpy(), g["ou_out"], rtol=1e-6, atol=1e-6)
    xi, fo = O.dense_estimator(P, "flow_estimators", T(g["de_x"]))
    np.testing.assert_allclose(xi.numpy(), g["de_xi"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(fo.numpy(), g["de_out"], rtol=1e-6, atol=1e-6)
    o = O.context_net(P, "context_networks", T(g["cn_x"]))
    np.testing.assert_allclose(o.numpy(), g["cn_out"], rtol=1e-6, atol=1e-6)


def test_param_inventory():
    P = O.synthetic_params(0)
    assert len(P) == 124
    assert sum(v.numel() for v in P.values()) == 6362092


@pytest.mark.parametrize("mode,thr", [("asis", 1.0), ("robust", 0.9999)])
def test_e2e_eval_bit_identical(golden_dir, mode, thr):
    g = _load(golden_dir, "e2e_B2_128x192.npz")
    P = O.synthetic_params(0)
    ck = np.array([float(sum(v.double().sum() for v in P.values())),
                   float(sum(v.double().abs().sum() for v in P.values()))])
    np.testing.assert_allclose(ck, g["weight_checksum"], rtol=0, atol=0)
    batch = O.synthetic_batch(2, 128, 192, 1234)
    with torch.no_grad():
        ev = O.irr_pwc_forward(P, batch["input1"], batch["input2"], False, mask_threshold=thr)
    # the restatement reproduces the imported reference exactly in eval mode
    epe = torch.norm(ev["flow"] - T(g[f"{mode}_eval_flow"]), dim=1).mean().item()
    assert epe <= 1e-6, epe
    assert (ev["occ"] - T(g[f"{mode}_eval_occ"])).abs().mean().item() <= 1e-6
    em = O.eval_metrics(ev, batch["target1"], batch["target_occ1"])
    np.testing.assert_allclose([float(em["epe"]), float(em["F1"])], g[f"{mode}_eval_metrics"], rtol=1e-6)


def test_e2e_train_robust(golden_dir):
    """loss / grad parity of one train step in robust-mask mode (as-is mode carries the
    reference's own mask self-noise, SURVEY.md finding 4, and is checked more loosely)."""
    g = _load(golden_dir, "e2e_B2_128x192.npz")
    names = [str(n) for n in g["param_names"]]
    for mode, thr, rtol in (("robust", 0.9999, 1e-4), ("asis", 1.0, 5e-2)):
        P = O.make_trainable(O.synthetic_params(0))
        opt = O.make_adam(P)
        batch = O.synthetic_batch(2, 128, 192, 1234)
        ld = O.train_step(P, opt, batch, mask_threshold=thr)
        got = np.array([ld["flow_loss"], ld["occ_loss"], ld["total_loss"]])
        np.testing.assert_allclose(got, g[f"{mode}_train_losses"], rtol=rtol if mode == "asis" else 1e-5)
        gn = np.array([float(P[n].grad.double().norm()) for n in names])
        ref = g[f"{mode}_train_gradnorm"]
        tot, tot_ref = np.sqrt((gn ** 2).sum()), np.sqrt((ref ** 2).sum())
        assert abs(tot - tot_ref) / tot_ref < rtol
        if mode == "robust":
            np.testing.assert_allclose(gn, ref, rtol=2e-3, atol=1e-4 * tot_ref)
            post = np.array([float(P[n].detach().double().sum()) for n in names])
            np.testing.assert_allclose(post, g["robust_poststep_sum"], rtol=1e-5, atol=1e-3)


def test_e2e_big_samples(golden_dir):
    g = _load(golden_dir, "e2e_B1_384x448.npz")
    P = O.synthetic_params(0)
    batch = O.synthetic_batch(1, 384, 448, 1234)
    idx = torch.from_numpy(g["sample_idx"])
    with torch.no_grad():
        ev = O.irr_pwc_forward(P, batch["input1"], batch["input2"], False, mask_threshold=0.9999)
    fl = ev["flow"].reshape(1, 2, -1)[:, :, idx]
    epe = torch.norm(fl - T(g["robust_flow_samples"]), dim=1).mean().item()
    assert epe <= 1e-4, epe


def test_e2e_train_B4_384x448_robust(golden_dir):
    """the oracle against the reference's train step at the size where the build's default routing uses the x3 kernels"""
    g = _load(golden_dir, "e2e_train_B4_384x448.npz")
    names = [str(n) for n in g["param_names"]]
    P = O.make_trainable(O.synthetic_params(0))
    opt = O.make_adam(P)
    ld = O.train_step(P, opt, O.synthetic_batch(4, 384, 448, 1234), mask_threshold=0.9999)
    np.testing.assert_allclose([ld["flow_loss"], ld["occ_loss"], ld["total_loss"]], g["robust_train_losses"], rtol=1e-5)
    gn = np.array([float(P[n].grad.double().norm()) for n in names])
    ref = g["robust_train_gradnorm"]
    tot_ref = np.sqrt((ref ** 2).sum())
    assert abs(np.sqrt((gn ** 2).sum()) - tot_ref) / tot_ref < 1e-4
    np.testing.assert_allclose(gn, ref, rtol=2e-3, atol=1e-4 * tot_ref)
    post = np.array([float(P[n].detach().double().sum()) for n in names])
    np.testing.assert_allclose(post, g["robust_poststep_sum"], rtol=1e-5, atol=1e-3)


def test_oddsize_vs_reference(golden_dir):
    """the oracle at sizes that are not multiples of 64 (odd pyramid sizes, half-pixel resize fallback, general adaptive
    pooling): Sintel-sized eval samples and a 100x132 train step against the imported reference"""
    g = _load(golden_dir, "oddsize.npz")
    P = O.synthetic_params(0)
    batch = O.synthetic_batch(1, 436, 1024, 1234)
    with torch.no_grad():
        ev = O.irr_pwc_forward(P, batch["input1"], batch["input2"], False, mask_threshold=0.9999)
    idx = torch.from_numpy(g["436x1024_idx"])
    epe = torch.norm(ev["flow"].reshape(1, 2, -1)[:, :, idx] - T(g["436x1024_flow_samples"]), dim=1).mean().item()
    assert epe <= 1e-5, epe
    names = [str(n) for n in g["param_names"]]
    Pt = O.make_trainable(O.synthetic_params(0))
    ld = O.train_step(Pt, O.make_adam(Pt), O.synthetic_batch(2, 100, 132, 1234), mask_threshold=0.9999)
    np.testing.assert_allclose([ld["flow_loss"], ld["occ_loss"], ld["total_loss"]], g["small_train_losses"], rtol=1e-5)
    gn = np.array([float(Pt[n].grad.double().norm()) for n in names])
    np.testing.assert_allclose(gn, g["small_train_gradnorm"], rtol=2e-3, atol=1e-4 * np.sqrt((g["small_train_gradnorm"] ** 2).sum()))
    post = np.array([float(Pt[n].detach().double().sum()) for n in names])
    np.testing.assert_allclose(post, g["small_poststep_sum"], rtol=1e-5, atol=1e-3)


def _oracle_three_steps(g, **adam_kw):
    names = [str(n) for n in g["param_names"]]
    P = O.make_trainable(O.synthetic_params(0))
    init = {n: P[n].detach().double().clone() for n in names}
    opt = O.make_adam(P) if not adam_kw else torch.optim.Adam(list(P.values()), **{**dict(lr=1e-4, weight_decay=4e-4), **adam_kw})
    losses, after1 = [], None
    for i, seed in enumerate(g["seeds"]):
        ld = O.train_step(P, opt, O.synthetic_batch(2, 128, 192, int(seed)), mask_threshold=0.9999)
        losses.append([ld["flow_loss"], ld["occ_loss"], ld["total_loss"]])
        if i == 0:
            after1 = {n: P[n].detach().double().clone() for n in names}
    d3 = {n: float((P[n].detach().double() - init[n]).norm()) for n in names}
    d31 = {n: float((P[n].detach().double() - after1[n]).norm()) for n in names}
    full = {str(n): (P[str(n)].detach().double() - init[str(n)]).numpy() for n in g["full_names"]}
    return losses, d3, d31, full


def test_three_optimizer_steps_vs_reference_and_checker_has_teeth(golden_dir):
    """Three Adam steps on three different batches: the oracle reproduces the imported reference (losses of all three steps,
    per-parameter update norms, small tensors element by element) -- and the SAME checker, at the SAME tolerances the GPU
    tests use, rejects an optimiser with a wrong beta2, beta1, weight decay, eps or learning rate."""
    from train3_check import problems
    g = _load(golden_dir, "train3_B2_128x192.npz")
    assert problems(g, *_oracle_three_steps(g)) == []
    for kw in (dict(betas=(0.9, 0.99)), dict(betas=(0.8, 0.999)), dict(weight_decay=0.0), dict(eps=1e-6), dict(lr=1.01e-4)):
        assert problems(g, *_oracle_three_steps(g, **kw)), f"checker accepted Adam with {kw}"


def test_e2e_train_448x1024_robust(golden_dir):
    """the oracle against the reference's train step at north_star's second crop (448x1024, BASELINE configs[4]), B = 1"""
    g = _load(golden_dir, "e2e_train_B1_448x1024.npz")
    names = [str(n) for n in g["param_names"]]
    P = O.make_trainable(O.synthetic_params(0))
    batch = O.synthetic_batch(1, 448, 1024, 1234)
    out = O.irr_pwc_forward(P, batch["input1"], batch["input2"], True, mask_threshold=0.9999)
    ld = O.multiscale_loss(out, batch["target1"], batch["target2"], batch["target_occ1"], batch["target_occ2"], batch_size=1)
    ld["total_loss"].backward()
    np.testing.assert_allclose([float(ld[k].detach()) for k in ("flow_loss", "occ_loss", "total_loss")], g["robust_train_losses"], rtol=1e-5)
    gn = np.array([float(P[n].grad.double().norm()) for n in names])
    ref = g["robust_train_gradnorm"]
    tot_ref = np.sqrt((ref ** 2).sum())
    assert abs(np.sqrt((gn ** 2).sum()) - tot_ref) / tot_ref < 1e-4
    np.testing.assert_allclose(gn, ref, rtol=2e-3, atol=1e-4 * tot_ref)
    np.testing.assert_allclose(out["flow"][4][2][:1, :, ::2, ::2].detach().numpy(), g["robust_train_l4_flow_f"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out["occ"][6][0][:1, :, ::8, ::8].detach().numpy(), g["robust_train_l6_occ_f"], rtol=1e-4, atol=1e-5)


def test_chunked_oracle_step_equals_whole_batch():
    """train_grads_chunked (what the GPU test of the full bs32 bench shape compares with, to bound host memory) == the
    whole-batch step: losses and every gradient."""
    batch = O.synthetic_batch(4, 128, 192, 5)
    P0 = O.make_trainable(O.synthetic_params(0))
    out = O.irr_pwc_forward(P0, batch["input1"], batch["input2"], True, mask_threshold=0.9999)
    ld = O.multiscale_loss(out, batch["target1"], batch["target2"], batch["target_occ1"], batch["target_occ2"], batch_size=4)
    ld["total_loss"].backward()
    P1 = O.make_trainable(O.synthetic_params(0))
    lc = O.train_grads_chunked(P1, batch, 2, mask_threshold=0.9999)
    for k in ("flow_loss", "occ_loss", "total_loss"):
        np.testing.assert_allclose(lc[k], float(ld[k].detach()), rtol=1e-6)
    tot = np.sqrt(sum(float((v.grad.double() ** 2).sum()) for v in P0.values()))
    for n in P0:
        d = float((P1[n].grad.double() - P0[n].grad.double()).norm())
        assert d <= 2e-4 * float(P0[n].grad.double().norm()) + 1e-6 * tot, (n, d)     # (fp32 summation order of the conv backward)


def test_general_correlation_restatement(golden_dir):
    """oracle.correlation_general (the legacy operator at any parameter point) against tests/golden/corr_general.npz: the imported
    reference's Python path at (md, 1, md, 1, 1) incl. gradients, the scalar transcription of correlation_cuda_kernel.cu:41-114
    elsewhere; and it IS cost_volume at the IRR-PWC point."""
    g = np.load(os.path.join(golden_dir, "corr_general.npz"))
    f1, f2 = torch.from_numpy(g["f1"]), torch.from_numpy(g["f2"])
    for md in (1, 2, 3):
        a, b = f1.clone().requires_grad_(True), f2.clone().requires_grad_(True)
        o = O.correlation_general(a, b, md, 1, md, 1, 1)
        o.backward(torch.from_numpy(g[f"ref_md{md}_go"]))
        np.testing.assert_allclose(o.detach().numpy(), g[f"ref_md{md}_out"], rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(a.grad.numpy(), g[f"ref_md{md}_g1"], rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(b.grad.numpy(), g[f"ref_md{md}_g2"], rtol=1e-12, atol=1e-14)
    for key in g.files:
        if key.startswith("scalar_"):
            pt = tuple(int(v) for v in key.split("_")[1:])
            np.testing.assert_allclose(O.correlation_general(f1, f2, *pt).numpy(), g[key], rtol=1e-12, atol=1e-14)
    assert torch.equal(O.correlation_general(f1, f2, 4, 1, 4, 1, 1), O.cost_volume(f1, f2, 4))
