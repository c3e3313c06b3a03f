# round 6's differential fuzz at builder scale: new seeds, the default 26-bit tables and -- in processes of their own -- 24- and 20-bit tables
# ($S2K_GTAB_BITS); the "more" half now carries the many-sums call and sums cut into several launches.
# tally -> gpurun_out/<tag>/fuzz_tally.txt:  bash tools/fuzz_round6.sh <tag>
cd ${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r06}; mkdir -p gpurun_out/$TAG; OUT=gpurun_out/$TAG/fuzz_tally.txt
: > $OUT
IFS=";" read -ra SPECS <<< "${S2K_FUZZ_SPECS:-601 26;602 26;603 26;604 24;605 20;606 20}"      # "seed bits;seed bits;..."
for spec in "${SPECS[@]}"; do
  set -- $spec
  for extra in "" more; do
    echo "seed $1 $extra, $2-bit tables" >> $OUT
    S2K_GTAB_BITS=$2 timeout 900 python tests/tools/fuzz_parity.py $1 1500 $extra 2>&1 | grep -i "mismatch\|error\|Traceback" >> $OUT
  done
done
echo "tallies: $(grep -c mismatches $OUT); with a mismatch: $(grep mismatches $OUT | grep -v 'mismatches: 0\|mismatches: \[\]' | wc -l)" | tee -a $OUT
