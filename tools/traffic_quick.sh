#!/bin/bash
# HBM fetch traffic of the scoring kernel (one PMC pass with FETCH_SIZE ALONE — combined with another
# counter the run hangs; KiB units, doubled as MI355X_MICROARCH.md prescribes for gfx950), kernel leg only.
# usage: bash tools/traffic_quick.sh <tag> [config]     (environment settings apply: SLG_LIB_TAG, SLG_INLINE_CUTS ...)
set -o pipefail
TAG=${1:-q}; CFG=${2:-c2}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/trafq_$TAG
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc -o run -- python3 $REPO/bench.py --config $CFG --steps 16 --warmup 2 --no-cpu-baseline --check 0 --kernel-leg-only > $OUT/pmc.log 2>&1
python3 - "$OUT" "$TAG" <<'PY'
import collections, csv, os, sys
acc = collections.defaultdict(list)
for r in csv.DictReader(open(os.path.join(sys.argv[1], "pmc", "run_counter_collection.csv"))):
    if "score_multi" in r["Kernel_Name"] or "score_uniform" in r["Kernel_Name"] or "partition" in r["Kernel_Name"]:
        acc[(r["Kernel_Name"].split("(")[0][-30:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k in sorted({k for k, _ in acc}):
    f = sum(acc[(k, "FETCH_SIZE")]) / len(acc[(k, "FETCH_SIZE")]) * 1024 * 2
    print(f"{sys.argv[2]:10s} {k:32s} fetch {f / 1e6:10.1f} MB per launch")
PY
