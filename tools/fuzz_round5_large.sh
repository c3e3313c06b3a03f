# builder-scale differential fuzz of the final round-5 library: 8 seeds x 6 000 items x 2 mixes (26-bit tables; two seeds on 22-bit tables)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r05L; OUT=gpurun_out/r05L/fuzz_tally.txt
: > $OUT
for spec in "701 26" "702 26" "703 26" "704 26" "705 26" "706 26" "707 22" "708 22"; do
  set -- $spec
  for extra in "" more; do
    echo "seed $1 $extra, $2-bit tables" >> $OUT
    S2K_GTAB_BITS=$2 timeout 1200 python tests/tools/fuzz_parity.py $1 6000 $extra 2>&1 | grep -i "mismatch\|error\|Traceback" >> $OUT
  done
done
echo "tallies: $(grep -c mismatches $OUT); with a mismatch: $(grep mismatches $OUT | grep -v 'mismatches: 0\|mismatches: \[\]' | wc -l)" | tee -a $OUT
