#!/bin/bash
# ablation timing of conv_x3 on the GPU box: rebuilds the library with -DX3_ABL=n and times the heavy shapes
for abl in 0 1 2 3; do
  echo "== X3_ABL=$abl"
  IRR_X3_ABL=$abl python -m irr_amd.build --force > /dev/null 2>&1
  python - <<'PY'
import torch, sys, os
sys.path.insert(0, os.getcwd())
from irr_amd import conv as C
from tools.test_x3 import timeit
for name, cin, cout, dil, B, H, W in [("ctx.conv0 L4", 565, 128, 1, 64, 96, 112), ("dense.conv5 L4", 531, 32, 1, 64, 96, 112), ("occup L6", 32, 32, 1, 32, 384, 448)]:
    x = torch.randn(B, cin, H, W, device="cuda"); w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05; b = torch.randn(cout, device="cuda")
    gf = 2.0 * B * H * W * cout * cin * 9 / 1e9
    t = timeit(lambda: C.conv_forward(x, w, b, 1, dil, True))
    print(f"{name:16s} {t:6.2f} ms {gf / t:6.1f} TF", flush=True)
PY
done
python -m irr_amd.build --force > /dev/null 2>&1
