#!/bin/bash
# Second fuzz campaign on the GPU box: two-level trees, deep trees, many lists
N=${1:-1500}
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
FUZZ_TREES=1 timeout -k 10 380 python tools/fuzz_parity.py $N 71 > gpurun_out/fuzz_long_trees.txt 2>&1; tail -n 2 gpurun_out/fuzz_long_trees.txt
FUZZ_DEEP=1 timeout -k 10 380 python tools/fuzz_parity.py $N 72 > gpurun_out/fuzz_long_deep.txt 2>&1; tail -n 2 gpurun_out/fuzz_long_deep.txt
FUZZ_MANY_LISTS=1 timeout -k 10 300 python tools/fuzz_parity.py 400 73 > gpurun_out/fuzz_long_many.txt 2>&1; tail -n 2 gpurun_out/fuzz_long_many.txt
