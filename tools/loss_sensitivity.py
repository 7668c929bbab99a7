"""Step-0 training loss (bs8, 384x448, MSRA init) under M=x3|f32 and with a relative input perturbation EPS (seed PSEED):
how much of a loss difference between kernel variants is rounding-level chaos of the model itself."""
import os, sys, types, torch
sys.path.insert(0, "/root/repo")
import irr_amd, bench
from irr_amd import conv as C
B = int(os.environ.get("SOAK_B", 8))
C.set_math(os.environ.get("M", "x3"))
args = types.SimpleNamespace(batch_size=B, model_div_flow=0.05)
torch.manual_seed(0)
model = irr_amd.PWCNet(args).cuda().train()
loss = irr_amd.MultiScaleEPE_PWC_Bi_Occ_upsample(args).train()
batch = bench.synthetic_batch(B, 384, 448, 1234, torch.device("cuda"))
eps = float(os.environ.get("EPS", 0))
if eps:
    g = torch.Generator(device="cuda").manual_seed(int(os.environ.get("PSEED", 1)))
    for k in ("input1", "input2"):
        batch[k] = batch[k] * (1 + eps * torch.randn(batch[k].shape, device="cuda", generator=g))
with torch.no_grad():
    out = model(batch)
    ld = loss(out, batch)
print(os.environ.get("TAG", ""), "total %.6f flow %.6f occ %.6f" % (float(ld["total_loss"]), float(ld["flow_loss"]), float(ld["occ_loss"])))
