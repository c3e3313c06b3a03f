R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/bpt
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/bpt -- python $R/tools/bppp_first_call.py 4096 16384 16384 > /tmp/bpt.out 2>/tmp/bpt.err
grep "n=" /tmp/bpt.out
f=$(find /tmp/bpt -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# split into phases by large gaps (> 200 ms) between dispatches
phases = [[]]
last = None
for r in rows:
    s = int(r['Start_Timestamp'])
    if last is not None and s - last > 150e6: phases.append([])
    phases[-1].append(r); last = int(r['End_Timestamp'])
for i, ph in enumerate(phases):
    agg = collections.defaultdict(list)
    for r in ph: agg[r['Kernel_Name'].split('(')[0][:36]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    tot = sum(sum(v) for v in agg.values())
    print("== phase %d: %d dispatches, %.1f ms of kernel time" % (i, len(ph), tot / 1e3))
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:7]:
        print("   %-38s n=%4d avg %9.1f us  min %9.1f  max %9.1f" % (k, len(v), sum(v) / len(v), min(v), max(v)))
PY
