R=${GRAFT_REPO_ROOT:-/root/repo}
for c in 6 7 8 9; do for d in 0 1 2 4 7; do echo "## c=$c dbg=$d"; S2K_LIB=$R/tools/ab_libs/lib_diag.so S2K_MM_C=$c S2K_MM_DEBUG=$d python $R/tools/msm_many_bare.py 256 1024 2>&1 | grep "K=" | cut -c1-70; done; done
