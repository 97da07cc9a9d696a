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
seg = corpus.zipf_segment(1_000_000, 1 << 18, seed=42)
offs, terms, w = corpus.zipf_queries(1024, int(os.environ.get("TERMS", "3")), seed=7, vocab=1 << 18)
ix = searcher.GpuIndex([seg])
b = ix.prepare(offs, terms, w, 11, int(os.environ.get("STRATEGY", "1")))
for _ in range(3):
    b.run()
b.sync()
info = b.info()
n = info["n_slices"]
out = np.zeros((n, 8), dtype=np.uint64)
L = N.load()
L.slg_debug_read_stamps.restype = C.c_int
L.slg_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
assert L.slg_debug_read_stamps(b._h, out.ctypes.data, n) == 0
names = ["0 plan/issue next", "1 chunk setup", "2 P0+P1 clear+or", "3 P2 read back", "4 P3 singles+queue",
         "5 P4 join", "6 wait loads+copy", "7 tail"]
ins = out[:, 7].copy(); out[:, 7] = 0
tot = out.sum()
print("inserts/slice mean", ins.mean(), "max", ins.max(), "p50", np.median(ins), "p90", np.percentile(ins, 90))
print("slices", n, "postings", info["n_postings"], "mean cycles/slice", tot / n)
for i, nm in enumerate(names):
    print(f"{nm:22s} {out[:, i].sum() / tot * 100:6.2f}%   mean/slice {out[:, i].mean():10.0f}")
