"""Debug helper (GPU box): one query of a keyed fuzz case against ONE segment; where the lost docs sit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import importlib.util
spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_parity.py"))
fz = importlib.util.module_from_spec(spec)
spec.loader.exec_module(fz)
import searchlite_amd as sa
from oracle import oracle as O
seed, it, qsel, si, nr = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
cap = {}
class Stop(Exception): pass
orig = sa.GpuIndex
class Fake:
    def __init__(self, segs, **kw): cap["segs"] = segs
    def __enter__(self): return self
    def __exit__(self, *a): return False
    def add_filter(self, masks): return 0
    def search_plan(self, offs, terms, w, k, **kw):
        cap["q"] = (offs, terms, w, k); raise Stop()
sa.GpuIndex = Fake
try: fz.run_case(seed, it)
except Stop: pass
sa.GpuIndex = orig
offs, terms, w, k = cap["q"]; segs = cap["segs"]
a, b = int(offs[qsel]), int(offs[qsel + 1])
o1 = np.array([0, b - a], dtype=np.uint32)
t1 = terms[a:b].copy(); w1 = w[a:b]
for s in range(t1.shape[1]):
    if s != si: t1[:, s] = 0xFFFFFFFF
import collections
for K, tuning in ((11, None), (11, {"rounds_per_slice": 8}), (11, {"rounds_per_slice": 1}), (11, {"rounds_per_slice": 2}),
                  (64, None), (65, None), (200, None), (11, {"inline_cuts": 1}), (11, {"uniform_round_target": 300})):
    want = O.search_batch(segs, o1, t1, w1, K, strategy=O.BM25)
    with sa.GpuIndex(segs, tuning=tuning) as ix:
        got = ix.search_batch(o1, t1, w1, K, 1)
    wl = [(int(want[0][0, r]), float(want[2][0, r])) for r in range(int(want[3][0]))]
    gl = [(int(got[0][0, r]), float(got[2][0, r])) for r in range(int(got[3][0]))]
    gd = {d for d, _ in gl}; wd = {d for d, _ in wl}
    dup = [d for d, c in collections.Counter(d for d, _ in gl).items() if c > 1]
    print("k", K, tuning, "missing", len([x for x in wl if x[0] not in gd]), "extra", len([x for x in gl if x[0] not in wd]), "dups", dup[:6], flush=True)
sys.exit(0)
seg = segs[si]
to = np.asarray(seg.term_offsets, dtype=np.int64)
tids = [int(x) for x in t1[:, si] if int(x) != 0xFFFFFFFF]
offs_l = [int(to[t]) + 64 * t for t in tids]; dfs_l = [int(to[t + 1] - to[t]) for t in tids]
lg = int(np.argmax(dfs_l))
P = int(to[-1]); V = len(to) - 1
docs = np.full(P + 64 * V + 576, 0xFFFFFFFF, dtype=np.uint64)
for t in range(V): docs[to[t] + 64 * t: to[t + 1] + 64 * t] = seg.doc_ids[to[t]:to[t + 1]]
docs8 = docs[::8]
b0 = [o >> 3 for o in offs_l]; nbl = [((o + d + 7) >> 3) - (o >> 3) for o, d in zip(offs_l, dfs_l)]
stride = (nbl[lg] + nr - 1) // nr
D = [0] + [int(docs8[b0[lg] + j * stride]) if j * stride < nbl[lg] and j < nr else 0xFFFFFFFF for j in range(1, nr + 1)]
for d, sc in [x for x in wl if x[0] not in gd][:4] + [x for x in gl if x[0] not in wd][:2]:
    r = max(j for j in range(nr) if D[j] <= d)
    print(f"doc {d} score {sc}: round {r} [{D[r]}, {D[r + 1]}) slice-round {r % 5}")
    for t, tid in enumerate(tids):
        lst = seg.doc_ids[to[tid]:to[tid + 1]]
        pos = int(np.searchsorted(lst, d))
        if pos < len(lst) and int(lst[pos]) == d:
            g = offs_l[t] + pos
            print(f"   list {t} (df {dfs_l[t]}{' splitter' if t == lg else ''}): posting {pos} rel block {g // 8 - b0[t]} entry {g % 8}")
