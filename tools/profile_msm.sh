# Counter evidence for the MSM kernels (separate rocprofv3 --pmc passes per counter set, --kernel-trace only beside them): memory-side traffic
# (FETCH_SIZE, WRITE_SIZE) and issue counters (SQ_*) of every kernel of one s2k_ecmult_multi_dev call at 2^20 and 2^24 terms.
#   usage: S2K_GIT_HEAD=<commit> bash tools/profile_msm.sh <tag> [sizes...]   -> gpurun_out/<tag>/msm_counters.json (stamped with the library's
#   sha256 and the sha256 of its sources, like the ring kernel's files: bench.py quotes msm.roofline.traffic / issued only from a matching file)
TAG=${1:-r05}; shift
SIZES=${@:-1048576 16777216}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
for N in $SIZES; do
  i=0
  for SET in "FETCH_SIZE" "WRITE_SIZE" \
             "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
             "GRBM_GUI_ACTIVE SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_FLAT" ; do
    i=$((i+1)); rm -rf $O/msm_${N}_p$i
    timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/msm_${N}_p$i -- python $R/tools/msm_kernel_breakdown.py $N > /dev/null 2>$O/msm_${N}_p$i.err || tail -3 $O/msm_${N}_p$i.err
  done
done
cd $R
python - "$TAG" $SIZES <<'PY'
import csv, glob, json, collections, sys, hashlib, os
tag = sys.argv[1]; sizes = [int(x) for x in sys.argv[2:]]
sys.path.insert(0, ".")
from secp256k1_zkp_amd import _native
stamp = {"so_sha256": hashlib.sha256(open("secp256k1_zkp_amd/libsecp256k1_zkp_amd.so", "rb").read()).hexdigest(), "src_sha256": _native.sources_sha256(),
         "git_head": os.environ.get("S2K_GIT_HEAD", "unknown")}
CALLS = 4                                   # tools/msm_kernel_breakdown.py makes four calls
out = dict(stamp); out["calls_per_run"] = CALLS
for n in sizes:
    per = {}
    for f in glob.glob("gpurun_out/%s/msm_%d_p*/**/*counter_collection.csv" % (tag, n), recursive=True):
        agg = collections.defaultdict(list); dur = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if not ("msm" in k or "gej" in k or "scan" in k): continue
            agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
            if "Start_Timestamp" in r: dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for (k, c), v in agg.items(): per.setdefault(k, {})[c] = {"launches": len(v), "sum_per_call": sum(v) / CALLS}
        for k, v in dur.items(): per.setdefault(k, {})["kernel_ns_under_profiler"] = {"launches": len(v), "sum_per_call": sum(v) / CALLS}
    tot = collections.defaultdict(float)
    for k, cs in per.items():
        for c, v in cs.items(): tot[c] += v["sum_per_call"]
    fetch_kb, write_kb = tot.get("FETCH_SIZE", 0.0), tot.get("WRITE_SIZE", 0.0)
    out[str(n)] = {"kernels": per, "per_call": {"hbm_bytes_raw": (fetch_kb + write_kb) * 1024, "hbm_bytes_fetch_x2": (2 * fetch_kb + write_kb) * 1024,
                                                 "valu_wave_instructions": tot.get("SQ_INSTS_VALU"), "int64_wave_instructions": tot.get("SQ_INSTS_VALU_INT64"),
                                                 "kernel_ns_under_profiler": tot.get("kernel_ns_under_profiler")},
                   "note": "sums over every kernel of one s2k_ecmult_multi_dev call (four calls per pass, divided by four); FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them; separate --pmc passes"}
    r1 = per.get("k_msm_round1", {})
    if "SQ_INSTS_VALU" in r1: print(n, "round 1: VALU wave-instructions per call", r1["SQ_INSTS_VALU"]["sum_per_call"], "per bucket addition (20 n, 16.4 n from 2^22):", r1["SQ_INSTS_VALU"]["sum_per_call"] * 64 / ((20 if n < (1 << 22) else 16.4) * n))
    print(n, json.dumps(out[str(n)]["per_call"]))
json.dump(out, open("gpurun_out/%s/msm_counters.json" % tag, "w"), indent=1)
PY
