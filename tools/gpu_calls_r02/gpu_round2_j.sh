#!/bin/bash
# Round-2 GPU call J: the rocprofv3 passes again (bench without the no-skip leg under the profiler)
set -o pipefail
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2j
bash tools/gpu_profile.sh r02 > gpurun_out/r2j/profile.log 2>&1; echo "profile rc=$?" | tee -a gpurun_out/r2j/status.txt
tail -5 gpurun_out/r2j/profile.log | cut -c1-300
cp profiles/r02_* profiles/traffic.json profiles/sweep_counters.json gpurun_out/r2j/ 2>/dev/null
tail -3 gpurun_out/prof/build_trace.log | cut -c1-200; tail -3 gpurun_out/prof/batch_trace.log | cut -c1-200
