#!/bin/bash
# Longer fuzz campaign on the GPU box (two flavours per call; a gpurun call is limited to 20 minutes)
# usage: bash tools/fuzz_long.sh <iterations> <seedA> <seedB>
N=${1:-2500}; SA=${2:-61}; SB=${3:-62}
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 560 python tools/fuzz_parity.py $N $SA > gpurun_out/fuzz_long_standard.txt 2>&1; tail -n 2 gpurun_out/fuzz_long_standard.txt
FUZZ_FEW_LISTS=1 timeout -k 10 560 python tools/fuzz_parity.py $N $SB > gpurun_out/fuzz_long_few_lists.txt 2>&1; tail -n 2 gpurun_out/fuzz_long_few_lists.txt
