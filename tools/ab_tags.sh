#!/bin/bash
# Same-box A/B of PREBUILT tagged libraries (no rebuild): tools/ab_tags.sh tagA tagB ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
ARGS=${AB_ARGS:-"--sizes 10000,30000,100000 --groups 4 --threads 1024 --sweeps 32"}
for round in 1 2 3; do
  for tag in "$@"; do
    echo "== round $round $tag"
    ASP_LIB_TAG=$tag ASP_NO_REBUILD=1 timeout -k 10 200 python tools/tune_sweep.py $ARGS | grep "M="
  done
done
