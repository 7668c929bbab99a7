#!/bin/bash
for abl in 0 1; do
  echo "== WX3_ABL=$abl"
  IRR_WX3_ABL=$abl python -m irr_amd.build --force > /dev/null 2>&1
  python tools/test_wx3.py 2>&1 | grep -E "^ctx.conv0 L4|^dense.conv1 L4|^refine 128->128|^occup 32->32 L6|^dense.conv4 L4"
done
python -m irr_amd.build --force > /dev/null 2>&1
