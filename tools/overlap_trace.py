"""How do the kernels of batches in flight overlap?  Reads a rocprofv3 --kernel-trace CSV of
`bench.py` and prints, for the scoring kernel: its duration, the time between consecutive starts,
and the fraction of its span during which ANOTHER scoring kernel is running too.
usage: python tools/overlap_trace.py <run_kernel_trace.csv>"""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
score = [(a, b) for a, b, n in rows if "score_" in n]
other = [(a, b, n) for a, b, n in rows if "score_" not in n and ("partition" in n or "merge" in n or "select" in n)]
if len(score) < 10:
    raise SystemExit("too few scoring kernels in the trace")
score = score[len(score) // 4:]  # skip warm-up
durs = [b - a for a, b in score]
gaps = [score[i + 1][0] - score[i][0] for i in range(len(score) - 1)]
ov = []
for i, (a, b) in enumerate(score):
    t = 0
    for j, (c, d) in enumerate(score):
        if i != j:
            t += max(0, min(b, d) - max(a, c))
    ov.append(t / (b - a))
med = lambda x: sorted(x)[len(x) // 2]
print(f"scoring kernels {len(score)}: median duration {med(durs) / 1e3:.1f} us, median start-to-start {med(gaps) / 1e3:.1f} us, "
      f"median overlap with other scoring kernels {med(ov) * 100:.0f} % of the span")
span = score[-1][1] - score[0][0]
busy = sum(durs)
print(f"span {span / 1e3:.0f} us, sum of scoring durations {busy / 1e3:.0f} us (ratio {busy / span:.2f}), per batch {span / len(score) / 1e3:.1f} us")
for name in ("partition", "merge"):
    d = [b - a for a, b, n in other if name in n]
    if d:
        print(f"{name}: {len(d)} launches, median {med(d) / 1e3:.1f} us")
