"""Debug helper (GPU box): one query of a keyed fuzz case under several tunings."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import importlib.util
spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_parity.py"))
fz = importlib.util.module_from_spec(spec)
spec.loader.exec_module(fz)
import tests.util as U
import searchlite_amd as sa
from oracle import oracle as O

seed, it, qsel = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
captured = {}
orig_index = sa.GpuIndex
class SpyIndex(orig_index):
    def __init__(self, segs, **kw):
        captured["segs"] = segs
        super().__init__(segs, **kw)
    def search_plan(self, offs, terms, w, k, **kw):
        captured["q"] = (offs, terms, w, k, kw)
        return super().search_plan(offs, terms, w, k, **kw)
sa.GpuIndex = SpyIndex
try:
    fz.run_case(seed, it)
except AssertionError:
    pass
sa.GpuIndex = orig_index
offs, terms, w, k, kw = captured["q"]
segs = captured["segs"]
a, b = int(offs[qsel]), int(offs[qsel + 1])
o1 = np.array([0, b - a], dtype=np.uint32)
t1, w1 = terms[a:b], w[a:b]
want = O.search_batch(segs, o1, t1, w1, k, strategy=O.BM25)
wset = [(int(want[1][0, r]), int(want[0][0, r])) for r in range(int(want[3][0]))]
def only(si):
    tt = t1.copy()
    for s in range(tt.shape[1]):
        if s != si:
            tt[:, s] = 0xFFFFFFFF
    return tt
for label, tq, tuning in (("both", t1, None), ("seg0 terms only", only(0), None), ("seg1 terms only", only(1), None),
                          ("both, natural order", t1, {"slice_order": 0}), ("both, again", t1, None),
                          ("both, sigma 1.0", t1, {"uniform_sigma_x100": 100})):
    want = O.search_batch(segs, o1, tq, w1, k, strategy=O.BM25)
    wset = [(int(want[1][0, r]), int(want[0][0, r])) for r in range(int(want[3][0]))]
    with sa.GpuIndex(segs, tuning=tuning) as ix:
        t1_ = tq
        bt = ix.prepare(o1, t1_, w1, k, 1)
        info = bt.info()
        bt.close()
        got = ix.search_batch(o1, t1_, w1, k, 1)
    gset = [(int(got[1][0, r]), int(got[0][0, r])) for r in range(int(got[3][0]))]
    miss = [x for x in wset if x not in gset]
    print(label, tuning, "slices", info.get("n_slices"), "missing", miss)
