"""Diagnostic: run one C2 batch with the SLG_STAMPS build and print per-phase cycle shares.
usage (GPU box): python tools/stamps.py [rounds_per_slice]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from searchlite_amd import build, corpus
build.GPU_LIB = build.build_gpu(stamps=True)  # (re)built when a source is newer than the library
from searchlite_amd import searcher, _native as N
if len(sys.argv) > 1:
    os.environ["SLG_ROUNDS_PER_SLICE"] = sys.argv[1]
# (DOCS=10000000 VOCAB=1048576 SEED=43 TERMS=5: config 3's corpus)
n_docs, vocab = int(os.environ.get("DOCS", "1000000")), int(os.environ.get("VOCAB", str(1 << 18)))
MF = int(os.environ.get("MF", "0"))  # MF=4: bench.py --config mf (4 fields, TERMS words per query string, a leaf per word)
if MF:
    seg = corpus.zipf_multifield_segment(n_docs, vocab, MF, seed=int(os.environ.get("SEED", "42")))
    offs, terms, w, leaf = corpus.multifield_queries(1024, int(os.environ.get("TERMS", "2")), MF, vocab, seed=7)
else:
    seg = corpus.zipf_segment(n_docs, vocab, seed=int(os.environ.get("SEED", "42")))
    offs, terms, w = corpus.zipf_queries(1024, int(os.environ.get("TERMS", "3")), seed=7, vocab=vocab)
    leaf = None
ix = searcher.GpuIndex([seg])
b = ix.prepare(offs, terms, w, int(os.environ.get("K", "11")), int(os.environ.get("STRATEGY", "1")), q_leaf=leaf)
for _ in range(3):
    b.run()
b.sync()
info = b.info()
n = info["n_slices"]
out = np.zeros((n, 12), dtype=np.uint64)
L = N.load()
L.slg_debug_read_stamps.restype = C.c_int
L.slg_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
assert L.slg_debug_read_stamps(b._h, out.ctypes.data, n) == 0
T_ = int(os.environ.get("TERMS", "3")) * (MF if MF else 1)
names = (["0 describe + issue next", "1 chunk setup / loop", "2 P0+P1 clear+or", "3 P2+P3 read back + flags", "4 queue build",
          "5 -", "6 join + candidates", "7 wait loads + settle"] if os.environ.get("SLG_UNIFORM_KERNEL", "3") != "2" else
         ["0 plan/issue next", "1 chunk setup", "2 P0+P1 clear+or", "3 P2 read back", "4 P3 queue",
          "5 singles", "6 P4 join", "7 wait loads+copy"]) if T_ <= int(os.environ.get("SLG_UNIFORM_MAX_TERMS", "8")) else \
        ["0 round setup (bounds, describe)", "1 P0 clear", "2 sweep A (bits)", "3 P2 prefix", "4 sweep C (accumulate)",
         "5 P4 top-k", "6 advance / tail", "7 -"]
ins = out[:, 8] & np.uint64(0xFFFFFFFF); queued = out[:, 8] >> np.uint64(32)
t0s, t1s, nr = out[:, 9].astype(np.int64), out[:, 10].astype(np.int64), (out[:, 11] >> np.uint64(32)).astype(np.int64)
xcc = (out[:, 11] & np.uint64(0xF)).astype(np.int64)
out = out[:, :8]
b0 = t0s.min()
t0s -= b0
t1s -= b0
span = t1s.max()
print("kernel span (10 ns ticks)", span, "sum of slice durations", int((t1s - t0s).sum()),
      "=> mean waves in flight", float((t1s - t0s).sum()) / span, "of", 256 * 4 * 6, "wave slots at 6/SIMD")
edges = np.linspace(0, span, 21)
for a, b in zip(edges[:-1], edges[1:]):
    mid = (a + b) / 2
    live = ((t0s <= mid) & (t1s > mid)).sum()
    started = ((t0s >= a) & (t0s < b)).sum()
    print(f"  t {a / span * 100:5.1f}%..{b / span * 100:5.1f}%: waves alive at mid {live:6d} started {started:6d}")
for r in sorted(set(nr.tolist())):
    m = nr == r
    print(f"  slices with {r:2d} rounds: {m.sum():6d}, mean duration {float((t1s - t0s)[m].mean()):10.0f}, per round {float((t1s - t0s)[m].mean()) / max(r, 1):8.0f}")
order = np.argsort(-(t1s - t0s))[:12]
print("longest slices: dur(10ns) start end rounds queued inserts | phase cycles 0..7")
for i in order:
    print(f"  {int(t1s[i] - t0s[i]):6d} {int(t0s[i]):6d} {int(t1s[i]):6d} {int(nr[i]):3d} {int(queued[i]):6d} {int(ins[i]):5d} | "
          + " ".join(f"{int(v):8d}" for v in out[i]))
late = np.argsort(-t1s)[:12]
print("last slices to finish:")
for i in late:
    print(f"  {int(t1s[i] - t0s[i]):6d} {int(t0s[i]):6d} {int(t1s[i]):6d} {int(nr[i]):3d} {int(queued[i]):6d} {int(ins[i]):5d} | "
          + " ".join(f"{int(v):8d}" for v in out[i]))
print("slices per XCC id", np.bincount(xcc, minlength=8).tolist())
print("queued postings (shared docs + aliases)", int(queued.sum()), "=", float(queued.sum()) / info["n_postings"] * 100, "% of postings")
tot = out.sum()
print("inserts/slice mean", ins.mean(), "max", ins.max(), "p50", np.median(ins), "p90", np.percentile(ins, 90))
print("slices", n, "postings", info["n_postings"], "mean cycles/slice", tot / n)
for i, nm in enumerate(names):
    print(f"{nm:22s} {out[:, i].sum() / tot * 100:6.2f}%   mean/slice {out[:, i].mean():10.0f}")
