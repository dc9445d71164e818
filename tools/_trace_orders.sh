export TMPDIR=/tmp ASP_LIB_TAG=abl ASP_NO_REBUILD=1 ASP_SHUFFLED_ABLATE=${ABL:-2}
cd /tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/abl2
timeout -k 10 400 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/abl2 -o t -- python3 $GRAFT_REPO_ROOT/tools/time_shuffled_big_batch.py ${N:-24} ${SW:-256} 2>&1 | grep "shuffled batch"
