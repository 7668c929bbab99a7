#!/bin/bash
# ablation timing of conv_wgrad_x3: tagged builds (irr_amd/lib_wx3abl<n>/), the product library is never touched
for abl in 0 1; do
  echo "== WX3_ABL=$abl"
  IRR_BUILD_TAG=wx3abl$abl IRR_WX3_ABL=$abl python -m irr_amd.build > /dev/null 2>&1
  IRR_HIP_LIB=irr_amd/lib_wx3abl$abl/libirr_hip.so python tools/wx3_check.py 2>&1 | grep -E "^ctx.conv0 L4|^dense.conv1 L4|^refine 128->128|^occup 32->32 L6|^dense.conv4 L4"
done
