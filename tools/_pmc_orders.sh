export TMPDIR=/tmp ASP_LIB_TAG=abl ASP_NO_REBUILD=1 ASP_SHUFFLED_ABLATE=2
OUT=$GRAFT_REPO_ROOT/gpurun_out/ordpmc
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 tools/time_shuffled_big_batch.py 24 64 > $OUT/fetch.log 2>&1 || { tail -5 $OUT/fetch.log; exit 3; }
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/l2 -- python3 tools/time_shuffled_big_batch.py 24 64 > $OUT/l2.log 2>&1 || { tail -5 $OUT/l2.log; exit 4; }
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- python3 tools/time_shuffled_big_batch.py 24 64 > $OUT/sq.log 2>&1 || { tail -5 $OUT/sq.log; exit 5; }
find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete
python3 tools/reduce_counters.py $OUT
