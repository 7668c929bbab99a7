"""Which `torch.empty` of the product path is read before it is written?  (round 5)

Every float32 HIP tensor that irr_amd allocates with torch.empty / torch.empty_like is filled with NaN (call sites are identified by
file:line inside irr_amd/); one train step (forward + loss + backward, weight-gradient lane on) then has to return the SAME loss and
gradients as the unpoisoned step, bit for bit on a single stream.  With all sites poisoned at once a leak shows as NaN; the sites are
then poisoned one at a time to name the allocation.  Found with it: see profiles/NOTES.md D.5.

usage: python3 tools/poison_probe.py [B H W] [--lane]"""
import os
import sys
import traceback
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import irr_amd  # noqa: E402
from irr_amd import ddp  # noqa: E402

argv = [a for a in sys.argv[1:] if not a.startswith("--")]
B, H, W = (int(argv[0]), int(argv[1]), int(argv[2])) if len(argv) >= 3 else (2, 128, 192)
PKG = os.path.dirname(os.path.abspath(irr_amd.__file__))

_empty, _empty_like = torch.empty, torch.empty_like
SITES = {}
MODE = {"on": False, "only": None, "value": float("nan")}


def _site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if fr.filename.startswith(PKG):
            return f"{os.path.basename(fr.filename)}:{fr.lineno}"
    return None


def _poison(t):
    if MODE["on"] and t.is_cuda and t.dtype == torch.float32 and t.numel() > 0:
        s = _site()
        if s is not None:
            SITES[s] = SITES.get(s, 0) + 1
            if MODE["only"] is None or MODE["only"] == s:
                t.fill_(MODE["value"])
    return t


def empty(*a, **k):
    return _poison(_empty(*a, **k))


def empty_like(*a, **k):
    return _poison(_empty_like(*a, **k))


torch.empty, torch.empty_like = empty, empty_like

args = types.SimpleNamespace(batch_size=B, model_div_flow=0.05)
torch.manual_seed(0)
model = irr_amd.PWCNet(args).cuda().train()
loss_mod = irr_amd.MultiScaleEPE_PWC_Bi_Occ_upsample(args).train()
arena = ddp.GradArena(model.named_parameters())
if "--lane" in sys.argv:
    arena.enable_async_wgrad()
batch = bench.synthetic_batch(B, H, W, 1234, torch.device("cuda"))


def step():
    arena.zero_grad()
    out = model(batch)
    ld = loss_mod(out, batch)
    ld["total_loss"].backward()
    arena.sync()
    torch.cuda.synchronize()
    return float(ld["total_loss"]), arena.flat.clone()


def _close(l, g):
    """the float atomics of the warp / bias kernels make two clean steps differ by ~1e-6 relative; a leak is NaN or gross"""
    return bool(torch.isfinite(g).all()) and abs(l - ref_loss) <= 1e-5 * abs(ref_loss) and \
        float((g - ref_grad).abs().max()) <= 1e-4 * float(ref_grad.abs().max())


step()
ref_loss, ref_grad = step()
print(f"{B} x {H} x {W}: reference loss {ref_loss!r}, |grad| {float(ref_grad.norm()):.6e}")
MODE["on"] = True
for value in (float("nan"), 1e30):
    MODE["value"], MODE["only"] = value, None
    SITES.clear()
    l, g = step()
    bad = int((~torch.isfinite(g)).sum())
    same = _close(l, g)
    print(f"poison {value}: {len(SITES)} allocation sites, {sum(SITES.values())} tensors; loss {l!r}, non-finite gradient values {bad}, "
          f"equal to the unpoisoned step (1e-4 of max |g|): {same}, max |diff| {float((g - ref_grad).abs().nan_to_num(nan=float('inf')).max()):.3e}")
    if same:
        continue
    for s in sorted(SITES):
        MODE["only"] = s
        l1, g1 = step()
        if not _close(l1, g1):
            print(f"   LEAK at {s}: loss {l1!r}, non-finite {int((~torch.isfinite(g1)).sum())}, "
                  f"max |diff| {float((g1 - ref_grad).abs().nan_to_num(nan=float('inf')).max()):.3e}")
