for t in "" scalar "" scalar; do
  if [ -z "$t" ]; then unset IRR_HIP_LIB; else export IRR_HIP_LIB=$PWD/irr_amd/lib_$t/libirr_hip.so; fi
  echo "== ${t:-product}"
  python tools/x3_check.py 2>&1 | grep -E "ctx.conv0 L4|dense.conv3 L4|dense.conv4 L4|dgrad ctx0|occup 32->32 L6"
  python tools/wx3_check.py 2>&1 | grep -E "ctx.conv0 L4|dense.conv4 L4|occup 32->32 L6"
  python tools/x3s_check.py 2>&1 | grep "occup L6 x3"
done
unset IRR_HIP_LIB
bash tools/ab_lib.sh scalar 3
