#!/bin/bash
# AddressSanitizer + UBSan over the two CPU libraries (the posting-file decoder of the product and
# the oracle): builds instrumented copies, swaps them in, runs the CPU tests that drive them, and
# restores the normal builds.  GPU sanitizers are not available on this pool.  usage: bash tools/sanitize_cpu.sh
set -e
cd "$(dirname "$0")/.."
T=$(mktemp -d)
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -shared -fPIC -Iinclude \
    -o $T/libslg_segfile.so searchlite_amd/csrc/slg_segfile.cpp
gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -shared -fPIC -pthread \
    -o $T/libslo_oracle.so oracle/slo_oracle.c -lm
cp searchlite_amd/lib/libslg_segfile.so $T/segfile.bak
cp oracle/libslo_oracle.so $T/oracle.bak
restore() { cp $T/segfile.bak searchlite_amd/lib/libslg_segfile.so; cp $T/oracle.bak oracle/libslo_oracle.so; }
trap restore EXIT
cp $T/libslg_segfile.so searchlite_amd/lib/libslg_segfile.so
cp $T/libslo_oracle.so oracle/libslo_oracle.so
touch searchlite_amd/lib/libslg_segfile.so oracle/libslo_oracle.so
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
  ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 \
  python -m pytest tests/test_segfile.py tests/test_oracle.py tests/test_golden.py tests/test_host.py -x -q -m "not gpu" -p no:cacheprovider
