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
