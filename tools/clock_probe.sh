cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02f; mkdir -p $O
timeout 400 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-msm > $O/bench_pmc.json 2>$O/pmc.err
cd $R
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r02f/pmc/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(list)
rows = list(csv.DictReader(open(f)))
print(rows[0].keys())
for r in rows:
    if r["Kernel_Name"].startswith("k_rp_rings"):
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in agg.items(): print(c, sum(v)/len(v), len(v))
kt = glob.glob("gpurun_out/r02f/pmc/**/*kernel_trace.csv", recursive=True)[0]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(kt)) if r["Kernel_Name"].startswith("k_rp_rings")]
print("durations ns", d)
PY
rocm-smi --showpower --showclocks 2>/dev/null | head -30
