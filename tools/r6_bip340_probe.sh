#!/bin/bash
# why is 2^16 BIP-340 sometimes 9x slower after the distinct-generator phase with the interleaved split tables?  kernel trace of one
# tools/ab_probe.py child per library
cd ${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp
T=${1:-r06ad}
for lib in tools/ab_libs/lib_r6_inplace.so secp256k1_zkp_amd/libsecp256k1_zkp_amd.so; do
  b=$(basename $lib .so)
  rm -rf gpurun_out/_p; 
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/_p -o t -- python tools/ab_probe.py $lib 1 > gpurun_out/${T}_bip340_probe_$b.txt 2>&1
  f=$(find gpurun_out/_p -name '*kernel_stats.csv' | head -1)
  echo "== $b" ; grep RESULT gpurun_out/${T}_bip340_probe_$b.txt
  python - "$f" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]: print("%-60s n=%5s avg %10.1f us  max %10.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["MaxNs"])/1e3))
P
done 2>&1 | tee gpurun_out/${T}_bip340_probe.txt
rm -rf gpurun_out/_p
