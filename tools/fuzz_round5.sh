# round 5's differential fuzz at builder scale: the default 26-bit tables and -- in processes of their own -- 20-bit tables ($S2K_GTAB_BITS)
# tally -> gpurun_out/<tag>/fuzz_tally.txt:  bash tools/fuzz_round5.sh <tag>
cd ${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r05}; mkdir -p gpurun_out/$TAG; OUT=gpurun_out/$TAG/fuzz_tally.txt
: > $OUT
for spec in "501 26" "502 26" "503 26" "504 20" "505 20"; do
  set -- $spec
  for extra in "" more; do
    echo "seed $1 $extra, $2-bit tables" >> $OUT
    S2K_GTAB_BITS=$2 timeout 900 python tests/tools/fuzz_parity.py $1 1500 $extra 2>&1 | grep -i "mismatch\|error\|Traceback" >> $OUT
  done
done
echo "tallies: $(grep -c mismatches $OUT); with a mismatch: $(grep mismatches $OUT | grep -v 'mismatches: 0\|mismatches: \[\]' | wc -l)" | tee -a $OUT
