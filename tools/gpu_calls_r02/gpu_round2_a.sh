#!/bin/bash
# Round-2 GPU call A: test-suite, issue-rate probe (plain + counter calibration), clean-exit check
# of the team profile under rocprofv3.
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2a
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/status.txt
hipcc --offload-arch=gfx950 -O3 tools/issue_rate_probe.hip -o /tmp/issue_rate_probe > $OUT/probe_build.log 2>&1 || exit 6
timeout -k 10 300 /tmp/issue_rate_probe > $OUT/issue_rate_probe.txt 2>&1; echo "probe rc=$?" | tee -a $OUT/status.txt
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/probe_pmc -- /tmp/issue_rate_probe > $OUT/probe_pmc.log 2>&1; echo "probe pmc rc=$?" | tee -a $OUT/status.txt
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/team_trace -- python tools/profile_team.py > $OUT/prof_team.log 2>&1; echo "team profile rc=$?" | tee -a $OUT/status.txt
tail -3 $OUT/pytest.log; cat $OUT/issue_rate_probe.txt; tail -5 $OUT/prof_team.log
