#!/bin/bash
# ablation timing of conv_x3 on the GPU box: builds tagged libraries with -DX3_ABL=n (irr_amd/lib_x3abl<n>/, the product
# library is never touched) and times the heavy shapes with each
for abl in 0 1 2 3; do
  echo "== X3_ABL=$abl"
  IRR_BUILD_TAG=x3abl$abl IRR_X3_ABL=$abl python -m irr_amd.build > /dev/null 2>&1
  IRR_HIP_LIB=irr_amd/lib_x3abl$abl/libirr_hip.so python - <<'PY'
import torch, sys, os
sys.path.insert(0, os.getcwd())
from irr_amd import conv as C
from tools.x3_check import timeit
for name, cin, cout, dil, B, H, W in [("ctx.conv0 L4", 565, 128, 1, 64, 96, 112), ("dense.conv5 L4", 531, 32, 1, 64, 96, 112), ("occup L6", 32, 32, 1, 32, 384, 448)]:
    x = torch.randn(B, cin, H, W, device="cuda"); w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05; b = torch.randn(cout, device="cuda")
    gf = 2.0 * B * H * W * cout * cin * 9 / 1e9
    t = timeit(lambda: C.conv_forward(x, w, b, 1, dil, True))
    print(f"{name:16s} {t:6.2f} ms {gf / t:6.1f} TF", flush=True)
PY
done
