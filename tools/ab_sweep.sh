#!/bin/bash
# Same-box A/B of sweep-kernel variants: tools/ab_sweep.sh "tagA:-DX=0" "tagB:-DX=1" ...
# Variants are built HERE-or-there with hipcc, then timed interleaved (3 rounds) on one GPU.
cd ${GRAFT_REPO_ROOT:-/root/repo}
ARGS=${AB_ARGS:-"--sizes 10000,100000 --groups 4 --threads 1024 --sweeps 32"}
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  ASP_LIB_TAG=$tag ASP_EXTRA_FLAGS="$flags" python -c "from annealing_sign_problem_amd import build; print(build.build())" || exit 1
done
for round in 1 2 3; do
  for spec in "$@"; do
    tag=${spec%%:*}; flags=${spec#*:}
    echo "== round $round variant $tag ($flags)"
    ASP_LIB_TAG=$tag ASP_EXTRA_FLAGS="$flags" timeout -k 10 200 python tools/tune_sweep.py $ARGS | grep "M="
  done
done
