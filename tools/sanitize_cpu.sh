#!/bin/bash
# AddressSanitizer + UBSan over the CPU libraries (the product's posting-file decoder, the product's
# host planner slg_plan.cpp behind its test C ABI, and the oracle): builds instrumented copies,
# swaps them in, runs the CPU tests that drive them, and restores the normal builds.  GPU sanitizers are not available on this pool.  usage: bash tools/sanitize_cpu.sh
set -e
cd "$(dirname "$0")/.."
T=$(mktemp -d)
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -shared -fPIC -Iinclude \
    -o $T/libslg_segfile.so searchlite_amd/csrc/slg_segfile.cpp
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -shared -fPIC -pthread \
    -o $T/libslg_plan.so searchlite_amd/csrc/slg_plan.cpp searchlite_amd/csrc/slg_plan_capi.cpp
gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -shared -fPIC -pthread \
    -o $T/libslo_oracle.so oracle/slo_oracle.c -lm
cp searchlite_amd/lib/libslg_segfile.so $T/segfile.bak
cp oracle/libslo_oracle.so $T/oracle.bak
python -c "from searchlite_amd import build; build.build_plan_lib()"
cp searchlite_amd/lib/libslg_plan.so $T/plan.bak
restore() { cp $T/segfile.bak searchlite_amd/lib/libslg_segfile.so; cp $T/oracle.bak oracle/libslo_oracle.so; cp $T/plan.bak searchlite_amd/lib/libslg_plan.so; }
trap restore EXIT
cp $T/libslg_segfile.so searchlite_amd/lib/libslg_segfile.so
cp $T/libslo_oracle.so oracle/libslo_oracle.so
cp $T/libslg_plan.so searchlite_amd/lib/libslg_plan.so
touch searchlite_amd/lib/libslg_segfile.so oracle/libslo_oracle.so searchlite_amd/lib/libslg_plan.so
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
  ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 \
  python -m pytest tests/test_segfile.py tests/test_oracle.py tests/test_golden.py tests/test_host.py tests/test_plan.py -x -q -m "not gpu" -p no:cacheprovider
