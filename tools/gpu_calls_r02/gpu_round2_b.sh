#!/bin/bash
# Round-2 GPU call B: which HIP feature breaks the exit under rocprofv3 (tools/exit_probe.hip)
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2b
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O2 tools/exit_probe.hip -o /tmp/exit_probe > $OUT/build.log 2>&1 || exit 6
for mode in plain fine coop; do
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/exit_$mode -- /tmp/exit_probe $mode > $OUT/exit_$mode.log 2>&1
  echo "exit_probe $mode under rocprofv3: rc=$?" | tee -a $OUT/status.txt
  timeout -k 10 60 /tmp/exit_probe $mode > $OUT/bare_$mode.log 2>&1
  echo "exit_probe $mode bare: rc=$?" | tee -a $OUT/status.txt
done
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py tests/test_gpu_published.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/status.txt
tail -3 $OUT/pytest.log
