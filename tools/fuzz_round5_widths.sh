cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r05w; OUT=gpurun_out/r05w/fuzz_tally.txt
: > $OUT
for spec in "601 26" "602 24" "603 22" "604 20" "605 26" "606 24"; do
  set -- $spec
  for extra in "" more; do
    echo "seed $1 $extra, $2-bit tables" >> $OUT
    S2K_GTAB_BITS=$2 timeout 900 python tests/tools/fuzz_parity.py $1 2000 $extra 2>&1 | grep -i "mismatch\|error\|Traceback" >> $OUT
  done
done
echo "tallies: $(grep -c mismatches $OUT); with a mismatch: $(grep mismatches $OUT | grep -v 'mismatches: 0\|mismatches: \[\]' | wc -l)" | tee -a $OUT
