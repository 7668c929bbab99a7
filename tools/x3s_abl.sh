#!/bin/bash
# ablation timing of conv_x3s: tagged builds (irr_amd/lib_x3sabl<n>/), the product library is never touched
for abl in 0 5 6 7; do
  echo "== X3_ABL=$abl"
  IRR_BUILD_TAG=x3sabl$abl IRR_X3_ABL=$abl python -m irr_amd.build > /dev/null 2>&1
  IRR_HIP_LIB=irr_amd/lib_x3sabl$abl/libirr_hip.so python - <<'PY'
import torch, sys, os
sys.path.insert(0, os.getcwd())
from irr_amd import conv as C
from tools.x3_check import timeit
for name, cin, cout, B, H, W in [("occup L6", 32, 32, 64, 384, 448)]:
    x = torch.randn(B, cin, H, W, device="cuda"); w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05; b = torch.randn(cout, device="cuda")
    t = timeit(lambda: C.conv_forward(x, w, b, 1, 1, True))
    print(f"{name:16s} {t:6.2f} ms", flush=True)
PY
done
