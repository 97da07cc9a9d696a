#!/bin/bash
# PMC comparison of library builds on one box (kernel leg only): instruction mix, wave / wait cycles, L2
# requests, HBM fetch — each counter set in its own pass, FETCH_SIZE alone (tools/traffic_quick.sh).
# usage: bash tools/pmc_ab.sh <config> <tag> [<tag> ...]     ("-" = the default build)
set -o pipefail
CFG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
ARGS="$REPO/bench.py --config $CFG --steps 8 --warmup 2 --no-cpu-baseline --check 0 --kernel-leg-only --coalesce-threads 0"
for t in "$@"; do
  if [ "$t" = "-" ]; then unset SLG_LIB_TAG; name=default; else export SLG_LIB_TAG=$t; name=$t; fi
  OUT=$REPO/gpurun_out/pmcab_${CFG}_$name
  rm -rf $OUT && mkdir -p $OUT
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --output-format csv -d $OUT/a -o run -- python3 $ARGS > $OUT/a.log 2>&1
  timeout -k 10 200 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -o run -- python3 $ARGS > $OUT/b.log 2>&1
  timeout -k 10 200 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum --output-format csv -d $OUT/c -o run -- python3 $ARGS > $OUT/c.log 2>&1
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/d -o run -- python3 $ARGS > $OUT/d.log 2>&1
  python3 - "$OUT" "$name" <<'PY'
import collections, csv, glob, os, sys
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "score_uniform" in r["Kernel_Name"] or "score_multi" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("==", sys.argv[2])
for c in sorted(acc):
    print(f"  {c:24s} {sum(acc[c]) / len(acc[c]):16.1f}  (n={len(acc[c])})")
PY
done
