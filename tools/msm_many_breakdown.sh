# per-kernel times of s2k_ecmult_multi_many_dev: bash tools/msm_many_breakdown.sh <tag> K n
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=$1; K=$2; N=$3
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/mmb
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mmb -- python $R/tools/msm_many_bare.py $K $N > /dev/null 2>/tmp/mmb.err
f=$(find /tmp/mmb -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $R/gpurun_out/${TAG}_msm_many_${K}x${N}_kernel_stats.csv || tail -5 /tmp/mmb.err
