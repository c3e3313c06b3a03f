# Memory-side and issue counters of k_rp_rings for both forms of the double multiplication (bench.py --rp-split 0/1 = S2K_OPT_RP_SPLIT; product builds read no S2K_RP_SPLIT from the environment): separate rocprofv3
# --pmc passes, kernel-trace only.   usage (GPU box): bash tools/profile_mem_counters.sh <tag>  -> gpurun_out/<tag>/mem_counters.json
TAG=${1:-r02mem}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
for SPLIT in 0 1; do
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_TOTAL_ACCESSES GRBM_GUI_ACTIVE" \
           "TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TCP_TCC_READ_REQ_LATENCY TCP_TCP_LATENCY TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_TA_TCP_STATE_READ" \
           "TCC_REQ TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL TCC_TAG_STALL TCC_BUSY" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/s${SPLIT}_pmc$i -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-msm --no-secondary --no-dropin --no-group --no-distinct --rp-split $SPLIT > /dev/null 2>$O/s${SPLIT}_pmc$i.err
  tail -1 $O/s${SPLIT}_pmc$i.err
done
done
cd $R
python - "$TAG" <<'PY'
import csv, glob, json, collections, sys, re
tag = sys.argv[1]
out = {}
for f in glob.glob("gpurun_out/%s/s*_pmc*/**/*counter_collection.csv" % tag, recursive=True):
    split = re.search(r"/s(\d)_pmc", f).group(1)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("k_rp_rings"): agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in sorted(agg.items()): out.setdefault("split=" + split, {})[c] = sum(v) / len(v)
json.dump(out, open("gpurun_out/%s/mem_counters.json" % tag, "w"), indent=1)
names = sorted(set(out.get("split=0", {})) | set(out.get("split=1", {})))
for n in names: print("%-34s %16.0f %16.0f" % (n, out.get("split=0", {}).get(n, float("nan")), out.get("split=1", {}).get(n, float("nan"))))
PY
