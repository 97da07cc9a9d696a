"""Host-side cost of the C ABI per 1024-query batch (config 2): plan + upload (slg_batch_prepare),
the full prepare -> run -> fetch -> destroy cycle, and the same cycle from several host threads."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from searchlite_amd import corpus, searcher
seg = corpus.zipf_segment(1_000_000, 1 << 18, seed=42)
offs, terms, w = corpus.zipf_queries(1024, 3, seed=7, vocab=1 << 18)
ix = searcher.GpuIndex([seg])
for _ in range(3):
    b = ix.prepare(offs, terms, w, 11); b.run(); b.sync(); b.close()
N = 40
t = time.perf_counter()
for _ in range(N):
    b = ix.prepare(offs, terms, w, 11)
    b.close()
t1 = (time.perf_counter() - t) / N
def cycle(n, stream=None):
    import torch
    s = torch.cuda.Stream()
    for _ in range(n):
        b = ix.prepare(offs, terms, w, 11); b.set_stream(s.cuda_stream); b.run(); b.fetch(); b.close()
t = time.perf_counter(); cycle(N); t2 = (time.perf_counter() - t) / N
print(f"prepare+destroy {t1*1e3:.3f} ms; prepare+run+fetch+destroy {t2*1e3:.3f} ms -> {1024/t2:.0f} q/s (1 host thread)")
for nth in (2, 4, 8):
    th = [threading.Thread(target=cycle, args=(N,)) for _ in range(nth)]
    t = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    dt = time.perf_counter() - t
    print(f"{nth} host threads: {1024 * N * nth / dt:.0f} q/s end to end (host buffers in, host buffers out)")
