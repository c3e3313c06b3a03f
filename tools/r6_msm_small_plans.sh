# Round 6: window width / path for 2^11 .. 2^13 terms (-DS2K_DIAG library)
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { echo "## $1"; shift; env "$@" python $R/tools/msm_bare.py 2048 3000 4096 6000 8192 12000 2>/dev/null | cut -c1-100; }
run "product" S2K_LIB=$R/secp256k1_zkp_amd/libsecp256k1_zkp_amd.so
for c in 9 10 11 12 13; do run "c=$c" S2K_LIB=$R/tools/ab_libs/lib_diag.so S2K_MSM_C=$c; run "c=$c, rounds path" S2K_LIB=$R/tools/ab_libs/lib_diag.so S2K_MSM_C=$c S2K_MSM_NO_SMALL=1; done
