#!/bin/bash
# GPU box, end of round 5: does the store guard (NOTES D.5) remove the two-rank shared-GPU NaN of round 4's arithmetic?
# Round 4's PLAIN low pieces (IRR_DEFS=-DH2_LO_UP_ON=0: 12 of 21 two-rank runs NaN, profiles/r5_nan_ab.txt) with and without the guard,
# alternating blocks of four runs on one box; then twenty two-rank runs of the product library.
#   IRR_BUILD_TAG=loplain   IRR_DEFS="-DH2_LO_UP_ON=0" python -m irr_amd.build
#   IRR_BUILD_TAG=loplainug IRR_DEFS="-DH2_LO_UP_ON=0 -DX3S_STORE_UNGUARDED=1" python -m irr_amd.build
rm -f gpurun_out/nan_hunt.txt
G=$PWD/irr_amd/lib_loplain/libirr_hip.so; U=$PWD/irr_amd/lib_loplainug/libirr_hip.so
bash tools/r5_nan_hunt.sh "plain low pieces, UNGUARDED stores (round 4's kernel):4:IRR_HIP_LIB=$U" "plain low pieces, guarded stores:4:IRR_HIP_LIB=$G" \
                          "plain low pieces, UNGUARDED stores, second block:4:IRR_HIP_LIB=$U" "plain low pieces, guarded stores, second block:4:IRR_HIP_LIB=$G" \
                          "product library:20:IRR_DUMMY=1" > /dev/null
grep -a "===\|^run" gpurun_out/nan_hunt.txt
