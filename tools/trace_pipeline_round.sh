#!/bin/bash
# rocprofv3 --kernel-trace of one annealing round of the kagome_36 pipeline (32 clusters); read the
# database with tools/trace_report.py gpurun_out/pipe_trace/t_results.db.  GPU box.
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
D=/tmp/k36; mkdir -p $D
H5=$D/heisenberg_kagome_36.h5
python3 -m annealing_sign_problem_amd.sector_ed --model heisenberg_kagome_36 --output $H5 --tol 1e-8 > $D/ed.log 2>&1 || { tail -5 $D/ed.log; exit 1; }
rm -rf gpurun_out/pipe_trace
ASP_PIPELINE_TIMING=1 timeout -k 10 500 rocprofv3 --kernel-trace -d gpurun_out/pipe_trace -o t -- python3 -m annealing_sign_problem_amd.sampled_components \
    --model heisenberg_kagome_36 --hdf5 $H5 --seed 435834 --order 2 --global-cutoff 1e-6 --jobs 16 \
    --number-samples 32 --output $D/out.csv --annealing --batch 32 --number-sweeps ${SW:-200} 2>&1 | grep pipeline
ls -la gpurun_out/pipe_trace
