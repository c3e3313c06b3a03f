# Round 6: window width sweep with the slice tail allowed at every width (-DS2K_DIAG library)
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r06}
OUT=$R/gpurun_out/${TAG}_msm_csweep.txt; : > $OUT
run() { echo "## $1" >> $OUT; shift; env "$@" python $R/tools/msm_bare.py $SIZES 2>/dev/null | cut -c1-140 >> $OUT; }
SIZES="${SIZES:-524288 1048576 2097152 4194304}"
run "product library" S2K_LIB=$R/secp256k1_zkp_amd/libsecp256k1_zkp_amd.so
for c in 13 14 15 16; do
  run "c=$c, slice tail" S2K_LIB=$R/tools/ab_libs/lib_diag.so S2K_MSM_C=$c S2K_MSM_SLICE_MAXC=16
  run "c=$c, old tail" S2K_LIB=$R/tools/ab_libs/lib_diag.so S2K_MSM_C=$c S2K_MSM_OLD_TAIL=1
done
