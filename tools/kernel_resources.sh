#!/bin/bash
# Register / scratch / LDS use of every kernel of one source file (compile only, no GPU needed):
#   tools/kernel_resources.sh annealing_sign_problem_amd/csrc/sa_shuffled.hip
cd "$(dirname "$0")/.." || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math \
  -I include -I annealing_sign_problem_amd/csrc -x hip -c "$1" -o /dev/null \
  -Rpass-analysis=kernel-resource-usage 2>&1 | c++filt | python3 -c '
import re, sys
name = None
row = {}
for line in sys.stdin:
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs|LDS Size \[bytes/block\]):\s+(.*?)\s*(\[-Rpass|$)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        if name:
            print(name, row)
        name, row = v, {}
    else:
        row[k.split(" [")[0]] = v
if name:
    print(name, row)
'
