# per-kernel times of the MSM at the given sizes: bash tools/msm_breakdown.sh <tag> <n>...   -> gpurun_out/<tag>_msm_<n>_kernel_stats.csv
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
  rm -rf /tmp/mb_$n
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mb_$n -- python $R/tools/msm_kernel_breakdown.py $n > /dev/null 2>/tmp/mb_$n.err
  f=$(find /tmp/mb_$n -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $R/gpurun_out/${TAG}_msm_${n}_kernel_stats.csv || tail -5 /tmp/mb_$n.err
done
