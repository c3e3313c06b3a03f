#!/bin/bash
# sums of fewer than 32 terms: the bucket-free path (one double multiplication per lane, the round-1 choice) against the bucket path
# (-DMSM_SMALL_N=1), same box; results compared by their first bytes
cd ${GRAFT_REPO_ROOT:-/root/repo}
for lib in tools/ab_libs/lib_r6_small32.so tools/ab_libs/lib_r6_small1.so; do
  echo "## $lib"
  S2K_LIB=$PWD/$lib timeout 300 python tools/msm_bare.py 1 2 3 5 8 16 31 32 33 64 2>&1 | grep "n="
done
