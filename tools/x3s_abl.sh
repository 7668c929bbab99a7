# same-box A/B of tagged conv_x3s builds: bash tools/x3s_abl.sh <tag> [<tag> ...]   ("" = product)
for t in "" "$@" "" "$@"; do
  if [ -z "$t" ]; then unset IRR_HIP_LIB; else export IRR_HIP_LIB=$PWD/irr_amd/lib_$t/libirr_hip.so; fi
  echo "== ${t:-product}"; python tools/x3s_check.py 2>&1 | grep "occup L. x3"
done
