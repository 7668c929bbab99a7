#!/bin/bash
# Ablation builds of the Winograd kernel (timing only): irr_amd/lib_wabl<n>/libirr_hip.so = the product objects + conv_wino.hip compiled
# with -DWINO_ABL=<n> (and any extra -D given after the list).   bash tools/wino_abl_build.sh "1 2 3 4 5" [extra defs]
set -e
cd "$(dirname "$0")/.."
python -m irr_amd.build > /dev/null
TAGSFX=${3:-}
for n in $1; do
  d=irr_amd/lib_wabl$n$TAGSFX
  mkdir -p $d
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-function -fno-slp-vectorize -fno-vectorize \
     -DWINO_ABL=$n $2 -c irr_amd/csrc/conv_wino.hip -o $d/conv_wino.o &
done
wait
for n in $1; do
  d=irr_amd/lib_wabl$n$TAGSFX
  objs=$(ls irr_amd/lib/obj/*.o | grep -v conv_wino.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libirr_hip.so $objs $d/conv_wino.o
  cp irr_amd/lib/source.hash $d/source.hash
done
ls -la irr_amd/lib_wabl*/libirr_hip.so
