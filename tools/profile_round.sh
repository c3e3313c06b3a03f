# One round's evidence on the GPU box: rocprofv3 kernel stats, HBM-side traffic (separate --pmc passes), SQ issue counters
# (separate --pmc passes), the plain default bench line, secondary throughputs and the MSM size sweep.
#   usage: S2K_GIT_HEAD=<commit> bash tools/profile_round.sh <tag>      -> gpurun_out/<tag>/...   (copy the summaries to profiles/<tag>_*)
set -x
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
# (the counter and stats passes time the shared-generator batch only: --no-distinct, --no-dropin; S2K_BENCH_PIPELINE=0 keeps one call in flight,
#  so that a kernel's duration is its own)
export S2K_BENCH_PIPELINE=0
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-distinct --no-dropin --no-msm-big --no-group --no-secondary --no-widths > $O/bench_under_rocprof.json 2>$O/stats.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-msm --no-distinct --no-dropin --no-group --no-secondary --no-widths > /dev/null 2>$O/fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-msm --no-distinct --no-dropin --no-group --no-secondary --no-widths > /dev/null 2>$O/write.err
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_FLAT" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/sq$i -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-msm --no-distinct --no-dropin --no-group --no-secondary --no-widths > /dev/null 2>$O/sq$i.err
done
unset S2K_BENCH_PIPELINE
cd $R
cp $(find gpurun_out/$TAG/stats -name "*kernel_stats.csv" | head -1) gpurun_out/$TAG/kernel_stats.csv
python - "$TAG" <<'PY'
import csv, glob, json, collections, sys, hashlib, subprocess
tag = sys.argv[1]
# every file written below is stamped with the commit and the sha256 of the library that produced the counters; bench.py quotes
# traffic / issued figures only from a file whose stamp matches the library it has loaded
so_sha = hashlib.sha256(open("secp256k1_zkp_amd/libsecp256k1_zkp_amd.so", "rb").read()).hexdigest()
import os
head = os.environ.get("S2K_GIT_HEAD", "unknown")      # the GPU box has no .git: pass it in,  S2K_GIT_HEAD=$(git rev-parse HEAD) bash tools/profile_round.sh <tag>
sys.path.insert(0, ".")
from secp256k1_zkp_amd import _native
stamp = {"so_sha256": so_sha, "src_sha256": _native.sources_sha256(), "git_head": head}
def collect(pattern, prefix="k_"):
    out = {}
    for f in glob.glob(pattern, recursive=True):
        agg = collections.defaultdict(list); dur = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
            if "Start_Timestamp" in r: dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for (k, c), v in sorted(agg.items()):
            if k.startswith(prefix): out.setdefault(k, {})[c] = {"launches": len(v), "mean_per_launch": sum(v) / len(v)}
        for k, v in dur.items():
            if k.startswith(prefix) and k in out: out[k]["kernel_ns_under_profiler"] = {"launches": len(v), "mean_per_launch": sum(v) / len(v)}
    return out
mem = collect("gpurun_out/%s/pmc_*/**/*counter_collection.csv" % tag)
rings = [k for k in ("k_rp_rings_shared", "k_rp_rings") if k in mem and "FETCH_SIZE" in mem[k] and "WRITE_SIZE" in mem[k]]
if rings:
    # both rings kernels run per launch group (the general one only exits where the shared-generator form served the wavefront): their sum
    f = sum(mem[k]["FETCH_SIZE"]["mean_per_launch"] for k in rings); w = sum(mem[k]["WRITE_SIZE"]["mean_per_launch"] for k in rings)
    json.dump({**stamp, "kernel": " + ".join(rings), "FETCH_SIZE_kb_per_launch": f, "WRITE_SIZE_kb_per_launch": w, "hbm_bytes_per_launch_raw": (f + w) * 1024,
               "hbm_bytes_per_launch_fetch_x2": (2 * f + w) * 1024,
               "note": "separate --pmc passes (MI355X_MICROARCH.md: FETCH_SIZE can read half of a wide stream on gfx950 -> the x2 figure is the upper bound); counters include Infinity-Cache hits; batch of 16384 proofs"},
              open("gpurun_out/%s/pmc_rp_rings.json" % tag, "w"), indent=1)
sq = collect("gpurun_out/%s/sq*/**/*counter_collection.csv" % tag)
json.dump({**stamp, **sq}, open("gpurun_out/%s/sq_counters.json" % tag, "w"), indent=1)
r = sq.get("k_rp_rings_shared", sq.get("k_rp_rings", {}))
if "GRBM_GUI_ACTIVE" in r and "kernel_ns_under_profiler" in r:
    print("effective clock GHz:", r["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8 / r["kernel_ns_under_profiler"]["mean_per_launch"])
print(json.dumps(r, indent=1)[:2500])
PY
timeout 600 python bench.py > gpurun_out/$TAG/bench_default.json 2> gpurun_out/$TAG/bench_default.err; tail -1 gpurun_out/$TAG/bench_default.json | cut -c1-600
timeout 600 python tests/tools/throughput.py > gpurun_out/$TAG/throughput.json 2> gpurun_out/$TAG/throughput.err; tail -5 gpurun_out/$TAG/throughput.json
timeout 300 python tests/tools/msm_sweep.py > gpurun_out/$TAG/msm_sweep.txt 2>&1; tail -20 gpurun_out/$TAG/msm_sweep.txt
# board power and shader clock under the sustained workload (sampled next to a separate 60-step run, never next to the bench line above)
(set +x; while true; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | sed -e 's/=//g' | tr "\n" " "; echo; sleep 0.25; done) > gpurun_out/$TAG/power_clock.txt 2>&1 &
SMI=$!
timeout 300 python bench.py --steps 200 --warmup 2 --no-cpu-baseline --no-msm --no-distinct --no-dropin --no-group --no-secondary --no-widths > gpurun_out/$TAG/bench_200_steps_with_smi_sampling.json 2>/dev/null
kill $SMI
grep -c "W" gpurun_out/$TAG/power_clock.txt; sort -t: -k5 gpurun_out/$TAG/power_clock.txt | tail -3
# the MSM alone: through the device entry point at a range of sizes, and its per-kernel times at 1 024, 2^20 and 2^24 terms
timeout 300 python tools/msm_bare.py 33 64 256 1024 4096 16384 65536 262144 1048576 4194304 16777216 > gpurun_out/$TAG/msm_bare.txt 2>/dev/null; cat gpurun_out/$TAG/msm_bare.txt
bash tools/msm_breakdown.sh $TAG/x 1024 1048576 16777216 2>/dev/null; ls gpurun_out/$TAG/
# round 5: counters of the MSM kernels (stamped like the files above)
S2K_GIT_HEAD=${S2K_GIT_HEAD:-unknown} bash tools/profile_msm.sh $TAG 1048576 16777216 2>&1 | tail -6
