# same-box A/B of two builds of the library: bash tools/ab_lib.sh <tag> [runs]   (tag = IRR_BUILD_TAG of the other build)
TAG=$1; N=${2:-3}
one() { python bench.py --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for i in $(seq $N); do
  echo -n "product: "; one
  echo -n "$TAG: "; IRR_HIP_LIB=$PWD/irr_amd/lib_$TAG/libirr_hip.so one
done
