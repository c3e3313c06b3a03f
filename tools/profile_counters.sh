# SQ / GRBM counter passes over bench.py's rangeproof step (separate rocprofv3 --pmc runs, kernel-trace only).
# usage (on the GPU box): bash tools/profile_counters.sh <tag>     -> gpurun_out/<tag>/counters_summary.json
set -x
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
rocprofv3 -L > $O/avail.txt 2>&1
grep -o "SQ_[A-Z0-9_]*\|GRBM_[A-Z0-9_]*" $O/avail.txt | sort -u > $O/avail_sq.txt
wc -l $O/avail_sq.txt
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC" \
           "GRBM_GUI_ACTIVE GRBM_COUNT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_IOPS SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_INSTS_FLAT" ; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/pmc$i -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-msm > /dev/null 2>$O/pmc$i.err
  tail -3 $O/pmc$i.err
done
cd $R
python - "$TAG" <<'PY'
import csv, glob, json, collections, sys
tag = sys.argv[1]
out = {}
for f in glob.glob("gpurun_out/%s/pmc*/**/*counter_collection.csv" % tag, recursive=True):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        if k.startswith("k_rp_"): out.setdefault(k, {})[c] = {"launches": len(v), "mean_per_launch": sum(v) / len(v)}
json.dump(out, open("gpurun_out/%s/counters_summary.json" % tag, "w"), indent=1)
print(json.dumps(out.get("k_rp_rings", {}), indent=1))
PY
