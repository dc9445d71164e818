#!/bin/bash
# Round-2 GPU call I: rocprofv3 passes (kernel trace, PMC, clock, build/batch traces), batch cap scan
set -o pipefail
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2i
timeout -k 10 600 python tools/tune_batch.py 128 512 > gpurun_out/r2i/tune_batch.log 2>&1; cat gpurun_out/r2i/tune_batch.log
bash tools/gpu_profile.sh r02 > gpurun_out/r2i/profile.log 2>&1; echo "profile rc=$?" | tee -a gpurun_out/r2i/status.txt
tail -5 gpurun_out/r2i/profile.log | cut -c1-300
cp profiles/r02_* profiles/traffic.json profiles/sweep_counters.json gpurun_out/r2i/ 2>/dev/null
ls gpurun_out/prof
