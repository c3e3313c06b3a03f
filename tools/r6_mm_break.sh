R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for c in ${CS:-7 8 9}; do
  rm -rf /tmp/mmb
  S2K_LIB=$R/tools/ab_libs/lib_diag.so S2K_MM_C=$c timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mmb -- python $R/tools/msm_many_bare.py ${KN:-256 1024} > /dev/null 2>/tmp/mmb.err
  f=$(find /tmp/mmb -name "*kernel_stats.csv" | head -1)
  echo "## c=$c"
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if True:
        print("%-32s calls %4s avg %10.1f us  min %10.1f  max %10.1f" % (r['Name'].split('(')[0][:32], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
done
