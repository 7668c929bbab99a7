"""which tensors does the per-channel scale of the weight gradient still measure with a pass (irr_amax_channels_f32), per step at the
BASELINE shape, and who asks -- the producers that fold their channel maxima on the way do not show up here"""
import os, sys, types, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import irr_amd
from irr_amd import conv as C, ddp
from irr_amd.train import ModelAndLoss
B, H, W = 32, 384, 448
args = types.SimpleNamespace(batch_size=B, model_div_flow=0.05)
torch.manual_seed(0)
m = irr_amd.PWCNet(args).cuda().train()
mal = ModelAndLoss(args, m, irr_amd.MultiScaleEPE_PWC_Bi_Occ_upsample(args)).train()
arena = ddp.GradArena(m.named_parameters())
if "--lane" in sys.argv:                                   # (default: weight gradients in line, so that the asking node is on the stack)
    arena.enable_async_wgrad()
batch = {"input1": torch.rand(B, 3, H, W, device="cuda"), "input2": torch.rand(B, 3, H, W, device="cuda"),
         "target1": torch.randn(B, 2, H, W, device="cuda"), "target2": torch.randn(B, 2, H, W, device="cuda"),
         "target_occ1": (torch.rand(B, 1, H, W, device="cuda") > 0.5).float(), "target_occ2": (torch.rand(B, 1, H, W, device="cuda") > 0.5).float()}
# who asks: the caller's layer (conv_wgrad: weight_shape / dil; conv_dgrad: weight / dil) beside every logged pass
_orig = C.channel_amax
LAYER = []
def _spy(t, out=None):
    f = sys._getframe(1).f_locals
    w = f.get("weight_shape") or (tuple(f["weight"].shape) if "weight" in f else None)
    if C.CHANNEL_PASS_LOG is not None:
        LAYER.append((f"{sys._getframe(1).f_code.co_name}", w, f.get("dil")))
    return _orig(t, out)
C.channel_amax = _spy
for it in range(2):
    if it == 1:
        C.CHANNEL_PASS_LOG = []
    arena.zero_grad()
    ld, _ = mal(batch)
    ld["total_loss"].backward()
    arena.sync()
    torch.cuda.synchronize()
agg = collections.Counter(); cnt = collections.Counter()
LAYER = LAYER[-len(C.CHANNEL_PASS_LOG):]
for (shape, stack), lay in zip(C.CHANNEL_PASS_LOG, LAYER):
    who = next((f for f in reversed(stack) if f.split(":")[0] not in ("channel_amax", "_spy", "conv_wgrad", "fn", "lazy", "launch", "_kick", "wgrad_param", "conv_dgrad")), "?")
    who = f"{who:14s} {lay[0]} weight {lay[1]} dil {lay[2]}"
    key = (who, shape)
    agg[key] += 4 * shape[0] * shape[1] * shape[2] * shape[3]; cnt[key] += 1
tot = sum(agg.values())
print(f"{len(C.CHANNEL_PASS_LOG)} passes per step, {tot / 1e9:.2f} GB read ({tot / 5.5e12 * 1e3:.2f} ms at 5.5 TB/s)")
for (who, shape), b in agg.most_common(25):
    print(f"  {b / 1e6:9.1f} MB  x{cnt[(who, shape)]:3d}  {str(shape):28s} {who}")
