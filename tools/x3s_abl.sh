for t in "" x3abl5 x3abl7 x3abl10 x3abl11; do
  if [ -z "$t" ]; then unset IRR_HIP_LIB; else export IRR_HIP_LIB=$PWD/irr_amd/lib_$t/libirr_hip.so; fi
  echo "== ${t:-product}"; python tools/x3s_check.py 2>&1 | grep "occup L. x3"
done
