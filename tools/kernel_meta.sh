#!/bin/bash
# usage: meta.sh [KREGS] [extra -D flags...]  -> register metadata of the uniform4 kernels
KR=${1:-1}; shift
cd /tmp/kmeta && rm -f *.s *.bc *.hipi *.out *.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DSLG_INST_KREGS=$KR "$@" --save-temps -c -o k.o /root/repo/searchlite_amd/csrc/slg_score_inst.hip 2>&1 | grep -E "error|warning: v" | head
awk '/\.name:/{name=$NF} /\.vgpr_count|vgpr_spill|sgpr_spill|private_segment_fixed|\.sgpr_count/{printf "%s %s %s\n", name, $1, $2}' slg_score_inst-hip-amdgcn-amd-amdhsa-gfx950.s | grep -E "uniform4" | sed 's/_ZN3slg21score_uniform4_kernelILi/u4<'/ | sed 's/EEEvNS_16RoundScoreParamsE//' | paste - - - - -
