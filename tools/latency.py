"""Latency of small calls through the C ABI (config-2 index): one query, 8 and 64 queries per call,
host buffers in and out (slg_batch_prepare -> run -> fetch -> destroy)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from searchlite_amd import corpus, searcher
seg = corpus.zipf_segment(1_000_000, 1 << 18, seed=42)
offs, terms, w = corpus.zipf_queries(1024, 3, seed=7, vocab=1 << 18)
ix = searcher.GpuIndex([seg])
for nq in (1, 8, 64, 1024):
    lat = []
    for rep in range(60):
        q0 = (rep * nq) % (1024 - nq + 1)
        o = (offs[q0:q0 + nq + 1] - offs[q0]).astype(np.uint32)
        t = terms[offs[q0]:offs[q0 + nq]]
        ww = w[offs[q0]:offs[q0 + nq]]
        t0 = time.perf_counter()
        b = ix.prepare(o, t, ww, 11); b.run(); r = b.fetch(); b.close()
        lat.append(time.perf_counter() - t0)
    lat = np.array(lat[10:]) * 1e6
    print(f"nq={nq:5d}: median {np.median(lat):8.1f} us  p90 {np.percentile(lat, 90):8.1f} us  "
          f"-> {nq / (np.median(lat) * 1e-6):10.0f} q/s per host thread")
