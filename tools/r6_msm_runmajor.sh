# Round 6, item "is round 1 memory co-limited at 2^24": bucket-major against run-major lane order of k_msm_round1 (-DS2K_DIAG library,
# $S2K_MSM_RUN_MAJOR = 1 / -1), timings and the memory-side counters of the kernel.   bash tools/r6_msm_runmajor.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r06}
OUT=$R/gpurun_out/${TAG}_msm_runmajor.txt; : > $OUT
for rm in -1 1; do
  echo "## S2K_MSM_RUN_MAJOR=$rm" >> $OUT
  S2K_LIB=$R/tools/ab_libs/lib_diag.so S2K_MSM_RUN_MAJOR=$rm python $R/tools/msm_bare.py 4194304 8388608 16777216 33554432 2>/dev/null | cut -c1-140 >> $OUT
done
cd /tmp && export TMPDIR=/tmp
for rm in -1 1; do
  i=0
  for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
             "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
             "FETCH_SIZE WRITE_SIZE"; do
    i=$((i+1)); rm -rf /tmp/rmj$i
    S2K_LIB=$R/tools/ab_libs/lib_diag.so S2K_MSM_RUN_MAJOR=$rm timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d /tmp/rmj$i -- python $R/tools/msm_kernel_breakdown.py 16777216 > /dev/null 2>/tmp/rmj$i.err || tail -3 /tmp/rmj$i.err
  done
  python - "$rm" >> $OUT <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/rmj*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("void k_msm_round1") or r["Kernel_Name"].startswith("k_msm_round1"): agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("## counters of k_msm_round1 at 2^24 terms, S2K_MSM_RUN_MAJOR=%s (average per launch)" % sys.argv[1])
for c, v in sorted(agg.items()): print("   %-28s %16.0f" % (c, sum(v) / len(v)))
PY
done
