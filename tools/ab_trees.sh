#!/bin/bash
# Whole bench line of two source trees on ONE box, alternating (the second tree: a built copy of an older
# commit under .r3tree/): usage: bash tools/ab_trees.sh <reps> [bench args...]
REPS=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in $(seq 1 $REPS); do for t in $R/.r3tree $R; do
  out=$(cd $t && python3 bench.py "$@" --no-cpu-baseline 2>/dev/null | tail -1)
  echo "tree=$(basename $t) $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("value", round(d["value"]), "ms_per_step", d["ms_per_step"], "spread", (d.get("value_spread") or {}).get("min"), (d.get("value_spread") or {}).get("max"), "resident", d["config"].get("kernel_only_qps"), "kernel_ms", d["roofline"]["kernel_ms"], "host", d["config"].get("host_call_ms"))' | cut -c1-330)"
done; done
