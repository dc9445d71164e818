#!/bin/bash
# One annealing round of the kagome_36 pipeline (32 clusters, shuffled order) per argument, each argument a
# list of environment settings — library variants (ASP_LIB_TAG=...), launch-shape knobs, or, with a
# -DASP_SHUF_ABLATE_ENV=1 build, ASP_SHUFFLED_ABLATE=1|2 (sweeps / orders alone; wrong results).  GPU box.
#   tools/pipeline_round_variants.sh "ASP_LIB_TAG=" "ASP_LIB_TAG=abl ASP_SHUFFLED_ABLATE=1"
export ASP_NO_REBUILD=1
cd ${GRAFT_REPO_ROOT:-/root/repo}
D=/tmp/k36; mkdir -p $D
H5=$D/heisenberg_kagome_36.h5
python3 -m annealing_sign_problem_amd.sector_ed --model heisenberg_kagome_36 --output $H5 --tol 1e-8 > $D/ed.log 2>&1 || { tail -5 $D/ed.log; exit 1; }
for cfg in "$@"; do
  echo "== $cfg"; rm -f $D/out.csv
  env $cfg ASP_PIPELINE_TIMING=1 timeout -k 10 200 python3 -m annealing_sign_problem_amd.sampled_components \
    --model heisenberg_kagome_36 --hdf5 $H5 --seed 435834 --order 2 --global-cutoff 1e-6 --jobs 16 \
    --number-samples 32 --output $D/out.csv --annealing --batch 32 2>&1 | grep "round of\|rror"
  md5sum < $D/out.csv
done
