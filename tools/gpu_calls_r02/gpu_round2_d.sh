#!/bin/bash
# Round-2 GPU call D: extended probe, batch tuning
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2d
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 tools/issue_rate_probe.hip -o /tmp/issue_rate_probe > $OUT/probe_build.log 2>&1 || exit 6
timeout -k 10 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/probe_pmc -- /tmp/issue_rate_probe > $OUT/probe_pmc.log 2>&1; echo "probe pmc rc=$?" | tee -a $OUT/status.txt
python tools/summarise_probe.py $OUT/probe_pmc r02 > $OUT/probe_summary.txt 2>&1
timeout -k 10 900 python tools/tune_batch.py 128 512 > $OUT/tune_batch.log 2>&1; echo "tune rc=$?" | tee -a $OUT/status.txt
cat $OUT/tune_batch.log
