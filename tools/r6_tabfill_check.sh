#!/bin/bash
# round 6: paired table fill (R + C and R - C from one denominator): parity of everything that reads a table, then the construction times
cd ${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp
T=${1:-r06ag}
timeout 900 python -m pytest tests/test_gpu_table_width.py tests/test_gpu_prims.py tests/test_gpu_kat.py tests/test_gpu_ecmult.py tests/test_gpu_bppp.py tests/test_gpu_gen_cache.py tests/test_gpu_rangeproof.py tests/test_gpu_pedersen.py -x -q < /dev/null 2>&1 | tail -3 > gpurun_out/${T}_tabfill_tests.txt
cat gpurun_out/${T}_tabfill_tests.txt
timeout 300 python tools/bppp_first_call.py < /dev/null 2>&1 | grep -v amdgpu.ids | head -4 > gpurun_out/${T}_tabfill_times.txt
rm -rf gpurun_out/_p
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/_p -o t -- python tools/bppp_first_call.py < /dev/null > /dev/null 2>&1
f=$(find gpurun_out/_p -name '*kernel_stats.csv' | head -1)
python - "$f" >> gpurun_out/${T}_tabfill_times.txt <<'P'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "tab" in r["Name"]: print("%-50s n=%3s avg %10.1f us" % (r["Name"][:50], r["Calls"], float(r["AverageNs"])/1e3))
P
rm -rf gpurun_out/_p
cat gpurun_out/${T}_tabfill_times.txt
