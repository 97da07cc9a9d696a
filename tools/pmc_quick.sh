#!/bin/bash
# Two PMC passes (instruction mix; wait / active cycles) of the scoring kernel, kernel leg only.
# usage: bash tools/pmc_quick.sh <tag> [config]   -> gpurun_out/pmcq_<tag>/summary.txt
set -o pipefail
TAG=${1:-q}; CFG=${2:-c2}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmcq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$REPO/bench.py --config $CFG --steps 16 --warmup 2 --no-cpu-baseline --check 0 --kernel-leg-only"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_a -o run -- python3 $ARGS > $OUT/pmc_a.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_b -o run -- python3 $ARGS > $OUT/pmc_b.log 2>&1
cd $REPO && python3 - "$OUT" > $OUT/summary.txt <<'PY'
import collections, csv, os, sys
d = sys.argv[1]
for p in ("pmc_a", "pmc_b"):
    path = os.path.join(d, p, "run_counter_collection.csv")
    if not os.path.exists(path):
        print(p, "missing"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if "score_multi" in r["Kernel_Name"] or "score_uniform" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[0][-28:], r["Counter_Name"])].append(float(r["Counter_Value"]))
            vg = r.get("VGPR_Count")
    for (k, c), v in sorted(acc.items()):
        print(f"{k:30s} {c:24s} {sum(v)/len(v):16.1f}  (n={len(v)})")
    print("VGPR_Count", vg)
PY
cat $OUT/summary.txt
