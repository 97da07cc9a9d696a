"""Host planning time (slg_batch_prepare incl. the descriptor upload) per batch, one caller thread.
usage (GPU box): python tools/prep_time.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from searchlite_amd import corpus, searcher

def run(name, segs, nq, T, k, vocab):
    offs, terms, w = corpus.zipf_queries(nq, T, seed=7, vocab=vocab)
    if len(segs) > 1:
        terms = np.repeat(terms.reshape(-1, 1), len(segs), axis=1).reshape(-1)
    with searcher.GpuIndex(segs) as ix:
        ts = []
        for _ in range(6):
            t0 = time.perf_counter()
            b = ix.prepare(offs, terms, w, k, searcher.Wand)
            t1 = time.perf_counter()
            info = b.info()
            b.close()
            ts.append((t1 - t0) * 1e3)
        print(f"{name}: prepare {min(ts):.2f} ms (median {sorted(ts)[3]:.2f}), slices {info['n_slices']}, postings {info['n_postings']}", flush=True)

s2 = corpus.zipf_segment(1_000_000, 1 << 18, seed=42, n_threads=16)
run("c2 (1024 x 3, 1M docs)", [s2], 1024, 3, 11, 1 << 18)
del s2
s4 = [corpus.zipf_segment(1_250_000, 1 << 20, seed=43 + r, n_threads=16) for r in range(int(os.environ.get("SEGS", "2")))]
run("c4 one rank of 8 (8192 x 5, 1 segment of 1.25M)", s4[:1], 8192, 5, 101, 1 << 20)
run(f"c4 {len(s4)} segments", s4, 8192, 5, 101, 1 << 20)
