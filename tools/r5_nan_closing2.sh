rm -f gpurun_out/nan_hunt.txt
G=$PWD/irr_amd/lib_loplain/libirr_hip.so; U=$PWD/irr_amd/lib_loplainug/libirr_hip.so
E="IRR_X3S_BITS=0 IRR_BRANCH_STREAMS=0"
bash tools/r5_nan_hunt.sh "plain low pieces, UNGUARDED, fp32 masks, one main stream (= the configuration of profiles/r5_nan_ab.txt):4:IRR_HIP_LIB=$U $E" "plain low pieces, guarded, same switches:4:IRR_HIP_LIB=$G $E" \
   "plain low pieces, UNGUARDED, same switches, second block:4:IRR_HIP_LIB=$U $E" "plain low pieces, guarded, same switches, second block:4:IRR_HIP_LIB=$G $E" > /dev/null
grep -a "===\|^run" gpurun_out/nan_hunt.txt
