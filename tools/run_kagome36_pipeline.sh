#!/bin/bash
# The sampled-cluster pipeline on the real 36-site kagome model, as separate COMMANDS on the GPU box
# (profiles/r0N_pipeline_commands.txt): the ground state once, then the runs given as arguments.
#   tools/run_kagome36_pipeline.sh "<samples> <extra flags>" ...     e.g. "64 --annealing --batch 32"
# Prints phase times (ASP_PIPELINE_TIMING), wall/user/sys and the md5 of every CSV.
cd ${GRAFT_REPO_ROOT:-/root/repo}
D=/tmp/k36; mkdir -p $D
H5=$D/heisenberg_kagome_36.h5
if [ ! -f $H5 ]; then
  python3 -m annealing_sign_problem_amd.sector_ed --model heisenberg_kagome_36 --output $H5 --tol 1e-8 > $D/ed.log 2>&1 || { tail -5 $D/ed.log; exit 1; }
  tail -1 $D/ed.log
fi
for run in "$@"; do
  set -- $run; samples=$1; shift
  rm -f $D/out.csv
  echo "== $samples clusters: $*"
  TIMEFORMAT="%R s wall, %U user, %S sys"
  time (ASP_PIPELINE_TIMING=1 python3 -m annealing_sign_problem_amd.sampled_components \
    --model heisenberg_kagome_36 --hdf5 $H5 --seed 435834 --order 2 --global-cutoff 1e-6 --jobs 16 \
    --number-samples $samples --output $D/out.csv "$@" 2>&1 | grep -v "amdgpu.ids")
  md5sum < $D/out.csv
done
