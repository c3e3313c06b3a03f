# Round 6: variants of the slice tail on ONE box (-DS2K_DIAG library, $S2K_MSM_SLICE_R = R + 8 * (64-lane workgroups)), next to the round-5 and the product library.
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r06}
SIZES="${SIZES:-1024 16384 131072 1048576}"
OUT=$R/gpurun_out/${TAG}_msm_variants.txt
mkdir -p $R/gpurun_out; : > $OUT
run() { echo "## $1" >> $OUT; shift; env "$@" python $R/tools/msm_bare.py $SIZES 2>/dev/null | cut -c1-140 >> $OUT; }
run "round-5 library" S2K_LIB=$R/tools/ab_libs/lib_r5.so
run "product library" S2K_LIB=$R/secp256k1_zkp_amd/libsecp256k1_zkp_amd.so
for v in ${VARIANTS:-1 2 4 9 10 12}; do run "diag library, S2K_MSM_SLICE_R=$v" S2K_LIB=$R/tools/ab_libs/lib_diag.so S2K_MSM_SLICE_R=$v; done
run "diag library, old tail" S2K_LIB=$R/tools/ab_libs/lib_diag.so S2K_MSM_OLD_TAIL=1
for l in ${LDS:-}; do run "diag library, S2K_MSM_SLICE_LDS=$l" S2K_LIB=$R/tools/ab_libs/lib_diag.so S2K_MSM_SLICE_LDS=$l; done
