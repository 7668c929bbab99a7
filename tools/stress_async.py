"""Stress the asynchronous weight-gradient lane: gradients of N repeated backward passes (async lane on) against the
synchronous single-stream gradients of the same inputs; prints every parameter whose gradient differs."""
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import irr_amd  # noqa: E402
from irr_amd import ddp  # noqa: E402
from irr_amd.train import ModelAndLoss  # noqa: E402
from oracle import irr_pwc_oracle as O  # noqa: E402

B, H, W = int(os.environ.get("SB", 2)), int(os.environ.get("SH", 128)), int(os.environ.get("SW", 192))
N = int(os.environ.get("SN", 20))
args = types.SimpleNamespace(batch_size=B, model_div_flow=0.05)
m = irr_amd.PWCNet(args, mask_threshold=0.9999)
m.load_state_dict(O.synthetic_params(0), strict=True)
m = m.cuda().train()
loss = irr_amd.MultiScaleEPE_PWC_Bi_Occ_upsample(args).train()
mal = ModelAndLoss(args, m, loss).train()
batch = {k: v.cuda() for k, v in O.synthetic_batch(B, H, W, 1234).items()}
arena = ddp.GradArena(m.named_parameters())
names = [n for n, _ in m.named_parameters()]


def grads(async_lane):
    if async_lane:
        arena.enable_async_wgrad()
    try:
        arena.zero_grad()
        ld, _ = mal(batch)
        ld["total_loss"].backward()
        arena.sync()
        torch.cuda.synchronize()
        return {n: p.grad.detach().clone() for n, p in m.named_parameters()}
    finally:
        if async_lane:
            arena.disable_async_wgrad()


m.branch_streams = False
ref = grads(False)
m.branch_streams = os.environ.get("IRR_BRANCH_STREAMS", "0") != "0"
print("branch streams:", m.branch_streams)
for mode in (False, True):
    bad = {}
    for it in range(N):
        g = grads(mode)
        for n in names:
            d = (g[n] - ref[n]).double().norm().item()
            r = ref[n].double().norm().item()
            if d > 1e-3 * r + 1e-7:
                bad.setdefault(n, []).append((it, d / max(r, 1e-30)))
    print("async" if mode else "sync", "lane:", "OK" if not bad else "")
    for n, v in bad.items():
        print("   ", n, tuple(ref[n].shape), ["it%d:%.2e" % x for x in v[:6]], "(%d of %d)" % (len(v), N))
