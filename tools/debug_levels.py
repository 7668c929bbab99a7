"""Compare irr_amd.PWCNet train-mode outputs with the oracle (CPU) level by level."""
import os, sys, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import irr_amd
from oracle import irr_pwc_oracle as O

thr = float(sys.argv[1]) if len(sys.argv) > 1 else 0.9999
B, H, W = 2, 128, 192
P = O.synthetic_params(0)
batch = O.synthetic_batch(B, H, W, 1234)
with torch.no_grad():
    ref = O.irr_pwc_forward(P, batch["input1"], batch["input2"], True, mask_threshold=thr)
m = irr_amd.PWCNet(types.SimpleNamespace(batch_size=B, model_div_flow=0.05), mask_threshold=thr)
m.load_state_dict(P); m = m.cuda().train()
with torch.no_grad():
    out = m({"input1": batch["input1"].cuda(), "input2": batch["input2"].cuda()})
for key in ("flow", "occ"):
    for l, (a, b) in enumerate(zip(out[key], ref[key])):
        d = [f"{(x.cpu() - y).abs().max().item():.2e}/{y.abs().max().item():.2e}" for x, y in zip(a, b)]
        print(key, "level", l, d)
m.eval()
with torch.no_grad():
    ev = m({"input1": batch["input1"].cuda(), "input2": batch["input2"].cuda()})
    rev = O.irr_pwc_forward(P, batch["input1"], batch["input2"], False, mask_threshold=thr)
print("eval flow max diff", (ev["flow"].cpu() - rev["flow"]).abs().max().item(), "EPE", torch.norm(ev["flow"].cpu() - rev["flow"], dim=1).mean().item())
print("eval occ max diff", (ev["occ"].cpu() - rev["occ"]).abs().max().item())
print("eval vs train l6*20", (ev["flow"].cpu() - out["flow"][6][0].cpu() * 20).abs().max().item())
import numpy as np
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/e2e_B2_128x192.npz"))
key = "robust" if thr < 1 else "asis"
print("oracle-eval vs golden EPE", torch.norm(rev["flow"] - torch.from_numpy(g[key + "_eval_flow"]), dim=1).mean().item())
print("build-eval vs golden EPE", torch.norm(ev["flow"].cpu() - torch.from_numpy(g[key + "_eval_flow"]), dim=1).mean().item())
