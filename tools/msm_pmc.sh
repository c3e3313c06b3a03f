# SQ counters of the MSM kernels at <n> terms (separate --pmc passes): bash tools/msm_pmc.sh <tag> <n>   -> gpurun_out/<tag>_msm_<n>_pmc.json
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=$1; N=$2
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU" \
           "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE" ; do
  i=$((i+1)); rm -rf /tmp/mp$i
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d /tmp/mp$i -- python $R/tools/msm_kernel_breakdown.py $N > /dev/null 2>/tmp/mp$i.err || tail -3 /tmp/mp$i.err
done
python - "$R/gpurun_out/${TAG}_msm_${N}_pmc.json" <<'PY'
import csv, glob, json, collections, sys
out = collections.defaultdict(dict)
for f in glob.glob("/tmp/mp*/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "msm" in k or "gej" in k: agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in agg.items(): out[k][c] = sum(v) / len(v)
json.dump(out, open(sys.argv[1], "w"), indent=1)
for k in ("void k_msm_bin<0>", "k_msm_round1", "k_msm_prep", "k_msm_bin_coarse", "k_msm_bin_fine"):
    if k in out: print(k, {c: round(v) for c, v in sorted(out[k].items())})
PY
