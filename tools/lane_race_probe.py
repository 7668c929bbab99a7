"""Lane vs single-stream gradients at 4 x 384x448, repeated (round 4, profiles/NOTES.md section C): the gradient with respect to the
images -- main-stream work only -- and every parameter gradient of N two-stream passes against one single-stream pass.
    python tools/lane_race_probe.py [N]            IRR_LANE_MAX_LEAD=0: unbounded lead (the state in which deviations were found)
    IRR_CONV_MATH=x3 ...                           the bf16x3 form (deviating passes are rare there)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_train_gpu import _setup, _batch
from irr_amd import conv as C

PB = int(os.environ.get("PB", "4"))                  # pairs per pass (round 5: PB=32 = the bench shape)
m, mal, arena, opt, step = _setup(PB, lane=False)
b = _batch(PB, 384, 448)


def grads(lane):
    if lane:
        arena.enable_async_wgrad()
    try:
        arena.zero_grad()
        for k in ("input1", "input2"):
            b[k].grad = None
            b[k].requires_grad_(True)
        ld, _ = mal(b)
        ld["total_loss"].backward()
        arena.sync()
        torch.cuda.synchronize()
        return arena.flat.clone(), torch.cat([b["input1"].grad.flatten(), b["input2"].grad.flatten()]).clone()
    finally:
        if lane:
            arena.disable_async_wgrad()


def rel(a, b_):
    return ((a - b_).double().norm() / (b_.double().norm() + 1e-300)).item()


ref, ref_img = grads(False)
from irr_amd import conv_nodes as _cn
REF_MAPS = None
if _cn._KEEP_LOG is not None:                                # the gradient maps of the single-stream pass, to compare the lane passes' with
    REF_MAPS = [(e[0], e[1].clone()) for e in _cn._KEEP_LOG if e[1].numel() >= 100_000_000][:12]
    _cn._KEEP_LOG.clear()
print("math", C.MATH, "IRR_LANE_MAX_LEAD", os.environ.get("IRR_LANE_MAX_LEAD", "1 (default)"), "IRR_LANE_GROUP", os.environ.get("IRR_LANE_GROUP", "4 (default)"))
for it in range(4):
    g, img = grads(False)
    if _cn._KEEP_LOG is not None:
        _cn._KEEP_LOG.clear()
    print(f"single stream {it}: image gradient {rel(img, ref_img):.2e}, parameters {rel(g, ref):.2e}", flush=True)
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    g, img = grads(True)
    off, worst = 0, (0.0, "")
    for n, p in m.named_parameters():
        k = p.numel()
        e = rel(g[off:off + k], ref[off:off + k])
        worst = max(worst, (e, n))
        off += k
    if _cn._KEEP_LOG is not None:                            # IRR_OCCUP_KEEP_LOG=1: which gradient map of the upsampler nodes went non-finite first
        for ent in _cn._KEEP_LOG:
            name, t = ent[0], ent[1]
            nf = int((~torch.isfinite(t)).sum())
            extra = ""
            if len(ent) > 2 and ent[2] is not None:          # the fused magnitude of the map against a pass over it
                mx = float(torch.where(torch.isfinite(t), t, torch.zeros_like(t)).abs().max())
                sl = float(ent[2][0][ent[2][1]])
                extra = f"; max |finite value| {mx:.6e}, fused slot {sl:.6e}" + ("  <-- slot BELOW the data" if sl < mx else "")
            print(f"      {name}: non-finite {nf} of {t.numel()}{extra}")
        if REF_MAPS is not None:
            big = [e for e in _cn._KEEP_LOG if e[1].numel() >= 100_000_000][:12]
            for (rn, rt), e in zip(REF_MAPS, big):
                d = (e[1] != rt) & ~(torch.isnan(e[1]) & torch.isnan(rt))
                nd = int(d.sum())
                if nd:
                    idx = d.flatten().nonzero().flatten()
                    B_, C_, H_, W_ = rt.shape
                    print(f"   DIFF {e[0]}: {nd} elements differ from the single-stream pass; first / last flat index {int(idx[0])} / {int(idx[-1])}")
                    for k in idx[:6].tolist() + idx[-2:].tolist():
                        b_, r = divmod(k, C_ * H_ * W_); c_, r = divmod(r, H_ * W_); y_, x_ = divmod(r, W_)
                        print(f"        (b {b_}, c {c_}, y {y_}, x {x_}): got {float(e[1].flatten()[k])!r}, single-stream {float(rt.flatten()[k])!r}")
                    bs = torch.zeros(B_, dtype=torch.long, device=d.device).scatter_add_(0, idx // (C_ * H_ * W_), torch.ones_like(idx))
                    cs = torch.zeros(C_, dtype=torch.long, device=d.device).scatter_add_(0, (idx // (H_ * W_)) % C_, torch.ones_like(idx))
                    print(f"        per sample: {bs.tolist()}\n        per channel: {cs.tolist()}")
                    break
        _cn._KEEP_LOG.clear()
    print(f"lane {it}: image gradient {rel(img, ref_img):.2e}, parameters {rel(g, ref):.2e}, worst parameter {worst[1]} {worst[0]:.1e}", flush=True)
