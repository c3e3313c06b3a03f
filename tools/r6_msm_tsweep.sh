# Round 6: run length of round 1 (T) at the middle sizes with the later rounds' run length chosen automatically (-DS2K_DIAG library)
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r06}
OUT=$R/gpurun_out/${TAG}_msm_tsweep.txt; : > $OUT
run() { echo "## $1" >> $OUT; shift; env "$@" python $R/tools/msm_bare.py $SIZES 2>/dev/null | cut -c1-140 >> $OUT; }
SIZES="${SIZES:-16384 65536 131072 262144 524288 1048576}"
run "product library" S2K_LIB=$R/secp256k1_zkp_amd/libsecp256k1_zkp_amd.so
for t in 8 12 16 24 32 48; do run "T=$t" S2K_LIB=$R/tools/ab_libs/lib_diag.so S2K_MSM_T=$t; done
for t2 in 4 5 6 8 10; do run "T2=$t2" S2K_LIB=$R/tools/ab_libs/lib_diag.so S2K_MSM_T2=$t2; done
