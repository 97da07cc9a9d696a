#!/bin/bash
# Fuzz parity on the final build (GPU box), four flavours; tails land in gpurun_out/fuzz_<name>.txt
# usage: bash tools/fuzz_final.sh [iterations per flavour]
N=${1:-600}
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 400 python tools/fuzz_parity.py $N 41 > gpurun_out/fuzz_standard.txt 2>&1; tail -n 2 gpurun_out/fuzz_standard.txt
FUZZ_FEW_LISTS=1 timeout -k 10 400 python tools/fuzz_parity.py $N 42 > gpurun_out/fuzz_few_lists.txt 2>&1; tail -n 2 gpurun_out/fuzz_few_lists.txt
FUZZ_TREES=1 timeout -k 10 400 python tools/fuzz_parity.py $N 43 > gpurun_out/fuzz_trees.txt 2>&1; tail -n 2 gpurun_out/fuzz_trees.txt
FUZZ_DEEP=1 timeout -k 10 400 python tools/fuzz_parity.py $N 44 > gpurun_out/fuzz_deep.txt 2>&1; tail -n 2 gpurun_out/fuzz_deep.txt
FUZZ_MANY_LISTS=1 timeout -k 10 300 python tools/fuzz_parity.py 150 45 > gpurun_out/fuzz_many_lists.txt 2>&1; tail -n 2 gpurun_out/fuzz_many_lists.txt
