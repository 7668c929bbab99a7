#!/bin/bash
# time the ablation builds of tools/wino_abl_build.sh on the gate shapes:  bash tools/wino_abl_run.sh "1 2 3 4 5"
export WINO_ONLY="${WINO_ONLY:-ctx.conv0 L4,dense.conv2 L4,dense.conv4 L4}"
echo "== product build"; python tools/wino_check.py --noacc 2>&1 | grep -v "^==\|amdgpu.ids"
for n in $1; do
  echo "== WINO_ABL=$n"; IRR_HIP_LIB=$PWD/irr_amd/lib_wabl$n/libirr_hip.so python tools/wino_check.py --noacc 2>&1 | grep -v "^==\|amdgpu.ids"
done
