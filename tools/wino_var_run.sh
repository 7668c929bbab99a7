#!/bin/bash
# time tagged builds of the Winograd kernel:  bash tools/wino_var_run.sh "0p0 0yo 2 4"   (irr_amd/lib_wabl<tag>/)
export WINO_ONLY="${WINO_ONLY:-ctx.conv0 L4,dense.conv2 L4,dense.conv4 L4}"
echo "== product build"; python tools/wino_check.py --noacc 2>&1 | grep -v "^==\|amdgpu.ids"
for n in $1; do
  echo "== variant $n"; IRR_HIP_LIB=$PWD/irr_amd/lib_wabl$n/libirr_hip.so python tools/wino_check.py --noacc 2>&1 | grep -v "^==\|amdgpu.ids"
done
if [ -d irr_amd/lib_wabl0tr ]; then echo "== trace"; IRR_HIP_LIB=$PWD/irr_amd/lib_wabl0tr/libirr_hip.so python tools/wino_trace.py 2>&1 | grep -v amdgpu.ids; fi
