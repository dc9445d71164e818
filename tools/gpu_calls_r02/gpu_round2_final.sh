#!/bin/bash
# Round-2 state check: full suite, smoke, bench as the driver runs it, kagome_36 pipeline vs host threads
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2final
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $OUT/status.txt
tail -5 $OUT/pytest.log | cut -c1-200
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; rc=$?; echo "smoke rc=$rc" | tee -a $OUT/status.txt; tail -1 $OUT/smoke.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.log 2>&1; rc=$?; echo "bench rc=$rc" | tee -a $OUT/status.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -m annealing_sign_problem_amd.sector_ed --model heisenberg_kagome_36 --output /tmp/k36.h5 > $OUT/ed.log 2>&1 || exit 5
for jobs in 1 4 8 16; do
  t0=$(date +%s%N)
  timeout -k 10 300 python -m annealing_sign_problem_amd.sampled_components --model heisenberg_kagome_36 --hdf5 /tmp/k36.h5 --seed 435834 \
    --output $OUT/k36_jobs$jobs.csv --order 2 --no-annealing --global-cutoff 1e-6 --number-samples 256 --jobs $jobs > $OUT/k36_jobs$jobs.log 2>&1 || exit 6
  echo "kagome_36 256 clusters x 3 orders greedy, --jobs $jobs: $(( ($(date +%s%N) - t0) / 1000000 )) ms (incl. reading the 504 MB ground-state file)" | tee -a $OUT/pipeline_jobs.txt
done
cmp $OUT/k36_jobs1.csv $OUT/k36_jobs8.csv && echo "outputs identical" | tee -a $OUT/pipeline_jobs.txt
tail -c 1200 $OUT/bench.log
