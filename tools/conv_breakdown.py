"""Per-shape time/TFLOPs of every conv launch (fwd, dgrad, wgrad) in one IRR-PWC train step."""
import os, sys, types, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import irr_amd
from irr_amd import conv as C, hip, ddp
from irr_amd.optim import FusedAdam
from irr_amd.train import ModelAndLoss, TrainStep
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 32
args = types.SimpleNamespace(batch_size=B, model_div_flow=0.05)
torch.manual_seed(0)
model = irr_amd.PWCNet(args).cuda().train()
loss = irr_amd.MultiScaleEPE_PWC_Bi_Occ_upsample(args).train()
arena = ddp.GradArena(model.named_parameters())
step = TrainStep(ModelAndLoss(args, model, loss), FusedAdam(model, arena), grad_sync=arena.sync)
batch = bench.synthetic_batch(B, 384, 448, 1234, torch.device("cuda"))
step(batch)
recs = []
orig = hip.call
def traced(name, *a):
    if name in ("irr_conv2d_fwd_f32", "irr_conv2d_wgrad_f32", "irr_conv2d_fwd_x3", "irr_conv2d_wgrad_x3", "irr_conv2d_wgrad_x3_dil"):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); orig(name, *a); e.record()
        if name == "irr_conv2d_fwd_x3":
            Bn, cin, H, W, cout, dil = a[5], a[6], a[7], a[8], a[9], a[10]
            key = ("x3 f/d", cin, cout, H, W, 3, dil)
            fl = 2.0 * Bn * H * W * cout * cin * 9
        elif name == "irr_conv2d_fwd_f32":
            Bn, cin, H, W, cout, oh, ow, k = a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12]
            key = ("fwd/dgrad", cin, cout, oh, ow, k, a[14])
            fl = 2.0 * Bn * oh * ow * cout * cin * k * k
        elif name == "irr_conv2d_wgrad_x3_dil":  # (x, gy, gw, ws, gbias, alpha, B, Cin, H, W, Cout, dil, ...)
            Bn, cin, H, W, cout, dil = a[6], a[7], a[8], a[9], a[10], a[11]
            key = ("x3 wgrad", cin, cout, H, W, 3, dil)
            fl = 2.0 * Bn * H * W * cout * cin * 9
        elif name == "irr_conv2d_wgrad_x3":      # (x, gy, gw, ws, gbias, alpha, B, Cin, H, W, Cout, ...)
            Bn, cin, H, W, cout = a[6], a[7], a[8], a[9], a[10]
            key = ("x3 wgrad", cin, cout, H, W, 3, 1)
            fl = 2.0 * Bn * H * W * cout * cin * 9
        else:                                    # (x, gy, gw, ws, gbias, alpha, B, Cin, H, W, Cout, OH, OW, k, stride, dil, ...)
            Bn, cin, H, W, cout, oh, ow, k = a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13]
            key = ("wgrad", cin, cout, oh, ow, k, a[15])
            fl = 2.0 * Bn * oh * ow * cout * cin * k * k
        recs.append((key, fl, s, e))
    else:
        orig(name, *a)
hip.call = traced
C.hip.call = traced
step(batch)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for key, fl, s, e in recs:
    a = agg.setdefault(key, [0, 0.0, 0.0]); a[0] += 1; a[1] += fl; a[2] += s.elapsed_time(e)
tot = sum(v[2] for v in agg.values())
print(f"total conv time {tot:.1f} ms, {sum(v[1] for v in agg.values())/1e12:.2f} TFLOP")
print(f"{'kind':10s} {'cin':>4s} {'cout':>4s} {'oh':>4s} {'ow':>4s} k dil {'n':>3s} {'ms':>8s} {'TF':>7s} {'%':>5s}")
for key, v in sorted(agg.items(), key=lambda kv: -kv[1][2])[:70]:
    print(f"{key[0]:10s} {key[1]:4d} {key[2]:4d} {key[3]:4d} {key[4]:4d} {key[5]} {key[6]:3d} {v[0]:3d} {v[2]:8.2f} {v[1]/v[2]/1e9:7.1f} {100*v[2]/tot:5.1f}")
if "--fp32" in sys.argv:
    print("every fp32-family launch shape:")
    t32 = 0.0
    for key, v in sorted(agg.items(), key=lambda kv: -kv[1][2]):
        if key[0] in ("fwd/dgrad", "wgrad"):
            t32 += v[2]
            print(f"{key[0]:10s} {key[1]:4d} {key[2]:4d} {key[3]:4d} {key[4]:4d} {key[5]} {key[6]:3d} {v[0]:3d} {v[2]:8.3f} {v[1]/v[2]/1e9:7.1f}")
    print(f"fp32 family total {t32:.2f} ms")
# by level
lev = collections.OrderedDict()
for key, v in agg.items():
    a = lev.setdefault((key[0], key[3], key[4]), [0.0, 0.0]); a[0] += v[1]; a[1] += v[2]
print("by level:")
for k, v in sorted(lev.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k[0]:10s} {k[1]:4d}x{k[2]:<4d} {v[1]:8.2f} ms {v[0]/v[1]/1e9:7.1f} TF")

lev = collections.OrderedDict()
for key, v in agg.items():
    a = lev.setdefault((key[3], key[0]), [0, 0.0, 0.0]); a[0] += v[0]; a[1] += v[1]; a[2] += v[2]
print("per level (output height) and kind:")
for (oh, kind), v in sorted(lev.items(), key=lambda kv: (-kv[0][0], kv[0][1])):
    print(f"  oh {oh:4d} {kind:10s} {v[0]:4d} launches {v[2]:8.2f} ms {v[1]/v[2]/1e9:7.1f} TF")
