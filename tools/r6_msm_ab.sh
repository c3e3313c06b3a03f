# Round 6: the MSM through the device entry point alone, round-5 library against the current one on ONE box, then per-kernel times.
#   bash tools/r6_msm_ab.sh <tag>     -> gpurun_out/<tag>_msm_ab.txt, gpurun_out/<tag>_msm_<n>_kernel_stats.csv
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r06}
SIZES="${SIZES:-64 1024 16384 131072 262144 1048576 4194304 16777216}"
OUT=$R/gpurun_out/${TAG}_msm_ab.txt
mkdir -p $R/gpurun_out; : > $OUT
for lib in $R/tools/ab_libs/lib_r5.so $R/secp256k1_zkp_amd/libsecp256k1_zkp_amd.so; do
  for rep in 1 2; do
    echo "## $(basename $lib) (run $rep)" >> $OUT
    S2K_LIB=$lib python $R/tools/msm_bare.py $SIZES 2>&1 | cut -c1-140 >> $OUT
  done
done
bash $R/tools/msm_breakdown.sh $TAG ${BREAK:-1024 131072 1048576}
