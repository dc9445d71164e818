#!/bin/bash
# Same-box A/B: the frozen older build libasp_hip_old.so vs the current libasp_hip.so
cd ${GRAFT_REPO_ROOT:-/root/repo}
ARGS=${AB_ARGS:-"--sizes 10000,30000,100000 --groups 4 --threads 1024 --sweeps 32"}
for round in 1 2 3; do
  echo "== round $round OLD"; ASP_LIB_TAG=old ASP_NO_REBUILD=1 timeout -k 10 200 python tools/tune_sweep.py $ARGS | grep "M="
  echo "== round $round NEW"; timeout -k 10 200 python tools/tune_sweep.py $ARGS | grep "M="
done
