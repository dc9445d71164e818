#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2u
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
PTS="heisenberg_kagome_16:400 heisenberg_kagome_16:3200 heisenberg_kagome_16:12800 sk_16_3:200 sk_16_3:12800 heisenberg_kagome_18:400 j1j2_square_4x4:100 sk_16_2:400 sk_16_1:400"
echo "== Metropolis (the specification)" > $OUT/rule.log
PROBE_LADDERS="geometric" timeout -k 10 700 python tools/schedule_probe.py $PTS >> $OUT/rule.log 2>&1
echo "== heat-bath / Glauber acceptance (experimental build)" >> $OUT/rule.log
ASP_LIB_TAG=glauber ASP_NO_REBUILD=1 PROBE_LADDERS="geometric" timeout -k 10 700 python tools/schedule_probe.py $PTS >> $OUT/rule.log 2>&1
grep -v amdgpu.ids $OUT/rule.log
