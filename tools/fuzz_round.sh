# the differential fuzzer at builder scale (6 seeds x 1 500 items x 2 mixes), tally -> gpurun_out/<tag>_fuzz_tally_large.txt:  bash tools/fuzz_round.sh <tag>
cd ${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03}
: > gpurun_out/${TAG}_fuzz_tally_large.txt
for seed in 101 102 103 104 105 106; do
  for extra in "" more; do
    echo "seed $seed $extra" >> gpurun_out/${TAG}_fuzz_tally_large.txt
    timeout 900 python tests/tools/fuzz_parity.py $seed 1500 $extra 2>&1 | grep -i "mismatch\|error\|Traceback" >> gpurun_out/${TAG}_fuzz_tally_large.txt
  done
done
grep -c "mismatches" gpurun_out/${TAG}_fuzz_tally_large.txt; grep "mismatches" gpurun_out/${TAG}_fuzz_tally_large.txt | grep -v "mismatches: 0\|mismatches: \[\]" | head
