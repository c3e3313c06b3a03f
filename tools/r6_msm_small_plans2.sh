R=${GRAFT_REPO_ROOT:-/root/repo}
run() { echo "## $1"; shift; env "$@" python $R/tools/msm_bare.py 100 300 512 1024 1500 2048 3000 4096 8192 12000 2>/dev/null | cut -c1-100; }
run "product" S2K_LIB=$R/secp256k1_zkp_amd/libsecp256k1_zkp_amd.so
for c in 7 8 9 10 11; do run "c=$c" S2K_LIB=$R/tools/ab_libs/lib_diag.so S2K_MSM_C=$c; done
