set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r01h
mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2>$O/stats.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>$O/fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>$O/write.err
cd $R
find gpurun_out/r01h -name "*.csv" | head -20
cp $(find gpurun_out/r01h/stats -name "*kernel_stats.csv" | head -1) gpurun_out/r01h/kernel_stats.csv
python - <<'PY'
import csv, glob, json, collections
out = {}
for tag in ("fetch", "write"):
    f = glob.glob("gpurun_out/r01h/pmc_%s/**/*counter_collection.csv" % tag, recursive=True)
    if not f: print("no counter file", tag); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        if k.startswith("k_"): out["%s:%s" % (k, c)] = {"launches": len(v), "mean_per_launch": sum(v) / len(v)}
json.dump(out, open("gpurun_out/r01h/pmc_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
timeout 400 python bench.py > gpurun_out/r01h/bench_default.json 2> gpurun_out/r01h/bench_default.err; tail -1 gpurun_out/r01h/bench_default.json | cut -c1-1500
