"""Block skipping on the query shape it is for: one stop-word-like term (rank 1..16) among four rare
terms, on config 3's corpus (10M docs), batch 4096, top-100.  Prints the scoring kernel's time and
the skipped share with block skipping on and off.   usage (GPU box): python tools/skewed_queries.py [docs]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from searchlite_amd import corpus, searcher

n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
vocab, nq, T, k = 1 << 20, 4096, 5, 101
seg = corpus.zipf_segment(n_docs, vocab, seed=43, n_threads=32)
rng = np.random.default_rng(5)
terms = np.empty((nq, T), dtype=np.uint32)
for q in range(nq):
    rare = rng.choice(np.arange(1024, 65536, dtype=np.uint32), size=T - 1, replace=False) - 1
    row = list(rare)
    row.insert(q % T, np.uint32(rng.integers(0, 16)))
    terms[q] = row
offs = (np.arange(nq + 1, dtype=np.uint32) * T).astype(np.uint32)
w = np.ones(nq * T, dtype=np.float32)
ref = None
variants = [{"block_max": 1}, {"block_max": 0}, {"pruning": 0}]
if os.environ.get("SKEW_VARIANTS"):  # e.g. "block_max=1,probe_target=3584;block_max=0"
    variants = [{kv.split("=")[0]: int(kv.split("=")[1]) for kv in v.split(",")} for v in os.environ["SKEW_VARIANTS"].split(";")]
for tuning in variants:
    with searcher.GpuIndex([seg], tuning=tuning) as ix:
        b = ix.prepare(offs, terms.reshape(-1), w, k, searcher.Wand)
        for _ in range(2):
            b.run()
        b.sync()
        ix.profile(True)
        for _ in range(5):
            b.run()
        b.sync()
        n, ms = ix.profile_read()
        got = b.fetch()
        info = b.info()
        probed, skipped = b.skip_counts()
        b.close()
    if ref is None:
        ref = got
    same = all(np.array_equal(a, c) for a, c in zip(got[:4], ref[:4]))
    print(f"tuning {tuning}: kernel {ms / n:.3f} ms, postings {info['n_postings']}, non-essential {probed}, "
          f"skipped {skipped} ({100.0 * skipped / max(info['n_postings'], 1):.1f} % of all postings), same hits {same}", flush=True)
