"""Diagnostic: keep references to the gradients every conv autograd node returns (no extra GPU work), single stream vs lane; list the
first nodes whose returned gradients deviate beyond the atomic-order noise."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_train_gpu import _setup, _batch
from irr_amd import conv as C, conv_nodes as N

REC = []
DGS = []
for cls in (N._ConvBlock, N._DenseEstimatorFn, N._ConvChainFn, N._OccUpsampleFn):
    orig = cls.backward

    def make(orig, name):
        def bw(ctx, *gs):
            out = orig(ctx, *gs)
            REC.append((name, [g for g in gs if g is not None], [o for o in out if isinstance(o, torch.Tensor)]))
            return out
        return staticmethod(bw)
    cls.backward = make(orig, cls.__name__)

DG = []
_orig_dgrad = N.conv_dgrad


def _dgrad(gy, weight, stride, dil, in_hw, **kw):
    out = _orig_dgrad(gy, weight, stride, dil, in_hw, **kw)
    DG.append(((tuple(gy.shape), tuple(weight.shape), dil, bool(kw.get("accumulate")), kw.get("gy_amax"), kw.get("gx_amax")), gy, out))
    return out


N.conv_dgrad = _dgrad
m, mal, arena, opt, step = _setup(4, lane=False)
b = _batch(4, 384, 448)


def run(lane):
    REC.clear()
    DG.clear()
    if lane:
        arena.enable_async_wgrad()
    try:
        arena.zero_grad()
        ld, _ = mal(b)
        ld["total_loss"].backward()
        arena.sync()
        torch.cuda.synchronize()
        DGS.append([(k, gy.clone(), o.clone(), (float(k[4].slots[k[4].first]) if k[4] is not None else None), (float(k[5].slots[k[5].first]) if k[5] is not None else None)) for k, gy, o in DG])
        return [(n, [t.clone() for t in i], [t.clone() for t in o]) for n, i, o in REC]
    finally:
        if lane:
            arena.disable_async_wgrad()


ref = run(False)
ref2 = run(False)


def rel(a, b_):
    return ((a - b_).double().norm() / (b_.double().norm() + 1e-300)).item()


def report(tag, cur):
    worst_in = max((rel(a, b_) for (_, i1, _), (_, i2, _) in zip(cur, ref) for a, b_ in zip(i1, i2)), default=0)
    print(f"{tag}: {len(cur)} nodes; worst incoming-gradient deviation {worst_in:.1e}")
    shown = 0
    for k, ((n1, i1, o1), (n2, i2, o2)) in enumerate(zip(cur, ref)):
        ein = max((rel(a, b_) for a, b_ in zip(i1, i2)), default=0)
        eout = max((rel(a, b_) for a, b_ in zip(o1, o2) if a.shape == b_.shape and a.dim() == 4), default=0)
        if eout > 3e-6 and shown < 8:
            shown += 1
            shapes = [tuple(t.shape) for t in o1 if t.dim() == 4]
            print(f"   node {k} {n1}: incoming deviation {ein:.1e} -> returned input-gradient deviation {eout:.1e}; returned {shapes}")
            if shown == 1:
                for a, b_ in zip(o1, o2):
                    if a.dim() != 4 or a.shape != b_.shape:
                        continue
                    d = (a - b_).abs()
                    per_b = [(d[i].double().norm() / (b_[i].double().norm() + 1e-300)).item() for i in range(a.shape[0])]
                    cg = max(1, a.shape[1] // 8)
                    per_c = [(d[:, c:c + cg].double().norm() / (b_[:, c:c + cg].double().norm() + 1e-300)).item() for c in range(0, a.shape[1], cg)]
                    rows = [(d[:, :, y0:y0 + 12].double().norm() / (b_[:, :, y0:y0 + 12].double().norm() + 1e-300)).item() for y0 in range(0, a.shape[2], 12)]
                    nz = int((d > 1e-3 * b_.abs().max()).sum())
                    print(f"      tensor {tuple(a.shape)}: per sample {[f'{v:.0e}' for v in per_b]}; per channel group of {cg} {[f'{v:.0e}' for v in per_c]}; per 12-row band {[f'{v:.0e}' for v in rows]}; elements off by > 1e-3 max: {nz}")


report("single stream again", ref2)
for it in range(6):
    report(f"lane run {it}", run(True))
    cur, base = DGS[-1], DGS[0]
    shown = 0
    for k, ((key, gy, o, sa, sb), (key0, gy0, o0, sa0, sb0)) in enumerate(zip(cur, base)):
        ei, eo = rel(gy, gy0), rel(o, o0)
        if eo > 3e-6 and shown < 6:
            shown += 1
            per_b = [f"{rel(o[i], o0[i]):.0e}" for i in range(o.shape[0])]
            print(f"      dgrad call {k} {key[:4]}: input deviation {ei:.1e} -> output {eo:.1e}; per sample {per_b}; gy_amax slot {sa} (single stream {sa0}), gx_amax slot {sb} (single stream {sb0})")
