#!/bin/bash
# CPU sanitizer pass over the host-side C++ of the library (plan layout incl. DSATUR, greedy
# tree): GPU AddressSanitizer is not available on the pool, the host code is where the pointer
# arithmetic lives.  Usage: bash tools/asan_host.sh   (build container, no GPU needed)
set -e
cd "$(dirname "$0")/.."
OUT=/tmp/asp_asan; mkdir -p $OUT
/opt/rocm/bin/hipcc -x c++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer \
  -D__HIP_PLATFORM_AMD__ -I /opt/rocm/include -I include -I annealing_sign_problem_amd/csrc -fPIC -shared \
  tools/asan_host/stub.cpp annealing_sign_problem_amd/csrc/sa_plan.cpp annealing_sign_problem_amd/csrc/greedy.cpp \
  -o $OUT/libhost_asan.so
LD_PRELOAD=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so) \
  ASAN_OPTIONS=detect_leaks=0 python tools/asan_host/run.py $OUT/libhost_asan.so
