# A/B of MSM launch parameters on ONE box: occupancy of round 1 (variant libraries built with -DMSM_R1_WAVES=n) and the run lengths of the
# partial-sum rounds ($S2K_MSM_T, $S2K_MSM_T2: -DS2K_DIAG builds read them at engine creation).   bash tools/msm_ab.sh > gpurun_out/<tag>_msm_ab.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { echo "## $1"; shift; env "$@" python $R/tools/msm_bare.py $SIZES 2>/dev/null | cut -c1-120; }
SIZES="131072 1048576 16777216"
for v in diag diag_w3 diag_w4; do run "library $v" S2K_LIB=$R/tools/ab_libs/lib_$v.so; done
SIZES="1048576"
for t2 in 8 10 12 16; do run "T2=$t2" S2K_LIB=$R/tools/ab_libs/lib_diag.so S2K_MSM_T2=$t2; done
for t in 16 20 32 40 48; do run "T=$t (T2 automatic)" S2K_LIB=$R/tools/ab_libs/lib_diag.so S2K_MSM_T=$t; done
SIZES="131072"
for t2 in 8 10 12 16; do run "T2=$t2" S2K_LIB=$R/tools/ab_libs/lib_diag.so S2K_MSM_T2=$t2; done
