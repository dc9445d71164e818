#!/bin/bash
# Runs on the GPU box: the rocprofv3 passes behind bench.py's `roofline` objects.  One run per case
# (tools/profile_roofline.py) and counter set, --kernel-trace only next to --pmc; then
# tools/summarise_roofline.py writes profiles/sweep_counters.json, profiles/traffic.json and the
# per-case kernel stats.  Usage: tools/gpu_profile_r4.sh [tag] [case ...]
set -o pipefail
export TMPDIR=/tmp
TAG=${1:-r04}; shift
CASES=${@:-"colour_10000 colour_30000 colour_100000 colour_200000 shuffled_10000 shuffled_30000 shuffled_100000 shuffled64_10000 batch batch_shuffled team real_kagome_36"}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof4
mkdir -p $OUT/cases
cd $GRAFT_REPO_ROOT
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
hipcc --offload-arch=gfx950 -O3 tools/fetch_calibrate.hip -o /tmp/fetch_calibrate > $OUT/calib_build.log 2>&1 || exit 6
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/calib -- /tmp/fetch_calibrate > $OUT/calib.log 2>&1 || exit 7
for c in $CASES; do
  echo "== $c"
  ARGS="tools/profile_roofline.py --case $c --out $OUT/cases"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$c/trace -- python3 $ARGS > $OUT/$c.trace.log 2>&1 || { tail -5 $OUT/$c.trace.log; exit 1; }
  rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $OUT/$c/sq -- python3 $ARGS > $OUT/$c.sq.log 2>&1 || { tail -5 $OUT/$c.sq.log; exit 2; }
  rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/$c/lanes -- python3 $ARGS > $OUT/$c.lanes.log 2>&1 || { tail -5 $OUT/$c.lanes.log; exit 5; }
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/$c/fetch -- python3 $ARGS > $OUT/$c.fetch.log 2>&1 || { tail -5 $OUT/$c.fetch.log; exit 3; }
  rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/$c/l2 -- python3 $ARGS > $OUT/$c.l2.log 2>&1 || { tail -5 $OUT/$c.l2.log; exit 4; }
done
# The box is fresh for every call and the cases take more than one call: the counter CSVs travel back
# (a few dispatches each, small) and tools/summarise_roofline.py runs where they have all arrived:
#   python tools/summarise_roofline.py gpurun_out/prof4 r04
find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete
python3 tools/reduce_counters.py $OUT
ls $OUT/cases
