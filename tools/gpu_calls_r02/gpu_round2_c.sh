#!/bin/bash
# Round-2 GPU call C: full test-suite (batched anneal included), clean-exit check of the team
# profile under rocprofv3 now that the team launch is an ordinary launch.
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/status.txt
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/team_trace -- python tools/profile_team.py > $OUT/prof_team.log 2>&1; echo "team profile under rocprofv3 rc=$?" | tee -a $OUT/status.txt
tail -5 $OUT/pytest.log; grep "team=" $OUT/prof_team.log
hipcc --offload-arch=gfx950 -O3 tools/issue_rate_probe.hip -o /tmp/issue_rate_probe > $OUT/probe_build.log 2>&1 || exit 6
timeout -k 10 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/probe_pmc -- /tmp/issue_rate_probe > $OUT/probe_pmc.log 2>&1; echo "probe pmc rc=$?" | tee -a $OUT/status.txt
python tools/summarise_probe.py $OUT/probe_pmc r02 > $OUT/probe_summary.txt 2>&1
timeout -k 10 900 python bench.py --steps 5 --warmup 2 > $OUT/bench.log 2>&1; echo "bench rc=$?" | tee -a $OUT/status.txt
tail -c 3000 $OUT/bench.log
