for t in "" x3abl12 "" x3abl12; do
  if [ -z "$t" ]; then unset IRR_HIP_LIB; else export IRR_HIP_LIB=$PWD/irr_amd/lib_$t/libirr_hip.so; fi
  echo "== ${t:-product}"
  python tools/x3_check.py 2>&1 | grep -E "ctx.conv0 L4|dense.conv1 L4|dense.conv3 L4|dense.conv4 L4|dgrad ctx0|ctx d2 L4|dense.conv2 L3"
done
