#!/bin/bash
# Runs on the GPU box: kernel-trace stats of bench.py, FETCH_SIZE calibration, then PMC
# passes (each in its own run, --kernel-trace only) on the sweep kernel.
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-build > $OUT/bench_trace.log 2>&1 || exit 1
hipcc --offload-arch=gfx950 -O3 tools/fetch_calibrate.hip -o /tmp/fetch_calibrate > $OUT/calib_build.log 2>&1 || exit 6
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_calib -- /tmp/fetch_calibrate > $OUT/pmc_calib.log 2>&1 || exit 7
# the SAME command as the bench line (rank 0, N=1), minus the CPU legs
ARGS="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-build"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $ARGS > $OUT/pmc_fetch.log 2>&1 || exit 2
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- python $ARGS > $OUT/pmc_l2.log 2>&1 || exit 3
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_sq1 -- python $ARGS > $OUT/pmc_sq1.log 2>&1 || exit 4
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc_sq2 -- python $ARGS > $OUT/pmc_sq2.log 2>&1 || exit 5
# the clock the sweep kernel really ran at (sum over the 8 XCDs / 8 / kernel time)
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_clock -- python $ARGS > $OUT/pmc_clock.log 2>&1 || exit 8
# the coupling-build / operator / sparsify kernels and the batched anneal: kernel-trace stats
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/build_trace -- python tools/profile_build.py > $OUT/build_trace.log 2>&1 || exit 9
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/batch_trace -- python tools/tune_batch.py 128 > $OUT/batch_trace.log 2>&1 || exit 10
python tools/summarise_profile.py $OUT ${1:-r02} > $OUT/summary.log 2>&1
cp profiles/${1:-r02}_* profiles/traffic.json profiles/sweep_counters.json $OUT/ 2>/dev/null
tail -5 $OUT/bench_trace.log | cut -c1-600
