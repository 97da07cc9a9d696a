"""Debug helper (GPU box): re-run one keyed fuzz case and, for the first differing query, print the
per-list contributions of the GPU's doc (numpy bm25), so a doubled or missing posting shows.
usage: FUZZ_FEW_LISTS=1 python tools/debug_case.py <seed> <iteration>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import importlib.util
spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_parity.py"))
fz = importlib.util.module_from_spec(spec)
spec.loader.exec_module(fz)
from tests.test_plan import impacts_of
import tests.util as U

seed, it = int(sys.argv[1]), int(sys.argv[2])
captured = {}
orig = U.assert_same_hits
def spy(got, want, tol, what):
    captured["got"], captured["want"] = got, want
    return orig(got, want, tol, what)
fz.assert_same_hits = spy
import searchlite_amd as sa
orig_index = sa.GpuIndex
class SpyIndex(orig_index):
    def __init__(self, segs, **kw):
        captured["segs"] = segs
        super().__init__(segs, **kw)
    def search_plan(self, offs, terms, w, k, **kw):
        captured["q"] = (offs, terms, w, k)
        return super().search_plan(offs, terms, w, k, **kw)
sa.GpuIndex = SpyIndex
try:
    fz.run_case(seed, it)
    print("case passes")
except AssertionError as e:
    print(str(e)[:300])
    got, want = captured["got"], captured["want"]
    offs, terms, w, k = captured["q"]
    segs = captured["segs"]
    for q in range(len(want[3])):
        n = int(want[3][q])
        i = 0
        while i < min(n, int(got[3][q])) and got[0][q, i] == want[0][q, i] and got[1][q, i] == want[1][q, i]:
            i += 1
        if i == n and int(got[3][q]) == n:
            continue
        gotset = {(int(got[1][q, r]), int(got[0][q, r])) for r in range(int(got[3][q]))}
        miss = [(int(want[1][q, r]), int(want[0][q, r])) for r in range(n) if (int(want[1][q, r]), int(want[0][q, r])) not in gotset]
        print("oracle docs missing from the GPU list:", miss[:4])
        import collections
        print("duplicates in the GPU list:", [d for d, c in collections.Counter((int(got[1][q, r]), int(got[0][q, r])) for r in range(int(got[3][q]))).items() if c > 1][:6])
        wantset = {(int(want[1][q, r]), int(want[0][q, r])) for r in range(n)}
        print("GPU docs the oracle does not have:", [x for x in gotset if x not in wantset][:4])
        dups = [d for d, c in collections.Counter((int(got[1][q, r]), int(got[0][q, r])) for r in range(int(got[3][q]))).items() if c > 1]
        gs, gd = dups[0] if dups else (miss[0] if miss else (int(got[1][q, i]), int(got[0][q, i])))
        seg = segs[gs]
        print("segment", gs, "n_docs", seg.n_docs)
        imp = impacts_of(seg)
        to = np.asarray(seg.term_offsets, dtype=np.int64)
        print(f"query {q} pos {i}: gpu doc (seg {gs}, {gd}) score {got[2][q, i]!r}")
        tot = np.float32(0)
        for j in range(int(offs[q]), int(offs[q + 1])):
            tid = int(terms[j, gs]) if terms.ndim == 2 else int(terms[j])
            if tid == 0xFFFFFFFF:
                print(f"  list {j - int(offs[q])}: absent term"); continue
            lo, hi = int(to[tid]), int(to[tid + 1])
            docs = seg.doc_ids[lo:hi]
            pos = int(np.searchsorted(docs, gd))
            if pos < len(docs) and int(docs[pos]) == gd:
                c = np.float32(imp[lo + pos]) * np.float32(w[j])
                tot = np.float32(tot + c)
                goff = lo + 64 * tid
                print(f"  list {j - int(offs[q])}: term {tid} df {hi - lo} global off {goff} (off%8={goff % 8}) posting #{pos} rel block {(goff + pos) // 8 - goff // 8} of {(goff + hi - lo + 7) // 8 - goff // 8} entry {(goff + pos) % 8}; docs around {docs[max(0, pos - 2):pos + 3].tolist()} contributes {c!r}")
            else:
                print(f"  list {j - int(offs[q])}: term {tid} df {hi - lo}: doc absent")
        print(f"  numpy sum {tot!r}")
        where = [(int(want[1][q, r]), int(want[0][q, r]), float(want[2][q, r])) for r in range(n) if int(want[0][q, r]) == gd and int(want[1][q, r]) == gs]
        print("  oracle has it as", where)
        break
