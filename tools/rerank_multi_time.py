"""Kernel time of the multi-clause hybrid rerank at config 5's shape (1M docs x 768-d f32 vectors,
1024 queries x 1001 random candidates -> top-10) for 1..8 clauses.
usage (GPU box): python tools/rerank_multi_time.py [dim] [metric 0|1]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from searchlite_amd import corpus, searcher, _native as N

dim = int(sys.argv[1]) if len(sys.argv) > 1 else 768
metric = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n_docs, nq, ncand, k_out = 1_000_000, 1024, 1001, 10
rng = np.random.default_rng(3)
big = corpus.zipf_segment(n_docs, 1 << 16, seed=42, n_threads=16)
vals = corpus.unit_vectors(n_docs, dim, seed=11)
L = N.load()
cd = torch.from_numpy(rng.integers(0, n_docs, size=(nq, ncand), dtype=np.int64).astype(np.int32)).cuda()
cs = torch.zeros((nq, ncand), dtype=torch.int32, device="cuda")
cb = torch.rand((nq, ncand), dtype=torch.float32, device="cuda")
cc = torch.full((nq,), ncand, dtype=torch.int32, device="cuda")
od = torch.empty((nq, k_out), dtype=torch.int32, device="cuda"); os_ = torch.empty_like(od)
osc = torch.empty((nq, k_out), dtype=torch.float32, device="cuda"); ov = torch.empty_like(osc)
oc = torch.empty((nq,), dtype=torch.int32, device="cuda")
for metric in ((0, 1) if len(sys.argv) <= 2 else (metric,)):
    big.vec_dim, big.vec_metric = dim, metric
    big.vec_offsets = np.arange(n_docs, dtype=np.uint32)
    big.vec_values = vals
    ix = searcher.GpuIndex([big])
    ix.set_stream(torch.cuda.current_stream().cuda_stream)
    print("metric", "cosine" if metric == 0 else "l2", flush=True)
    for nc in (1, 2, 4, 8):
        qv = torch.from_numpy(corpus.unit_vectors(nq * nc, dim, seed=12).reshape(nq, nc, dim)).cuda()
        al = torch.full((nq, nc), 0.5, dtype=torch.float32, device="cuda")
        bo = torch.full((nq, nc), 1.25, dtype=torch.float32, device="cuda")  # a boost: the multi-clause kernel also for nc = 1
        def run():
            N.check(L.slg_rerank_multi_batch_device(ix._h, nq, nc, qv.data_ptr(), al.data_ptr(), bo.data_ptr(),
                                                    cd.data_ptr(), cs.data_ptr(), cb.data_ptr(), cc.data_ptr(), ncand, k_out,
                                                    od.data_ptr(), os_.data_ptr(), osc.data_ptr(), ov.data_ptr(), oc.data_ptr()))
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            run()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 5
        gb = nq * ncand * dim * 4 / 1e9
        print(f"clauses {nc}: {ms:.3f} ms, row bytes {gb:.2f} GB -> {gb / ms:.2f} TB/s", flush=True)
    # clauses over DIFFERENT vector fields (rerank_fields_kernel): field 0 = the segment's own store,
    # field 1 = a second 384-d store of the other metric; 2 and 4 clauses alternating between them
    d2 = 384
    vals2 = corpus.unit_vectors(n_docs, d2, seed=31)
    f1 = ix.add_vector_field([(1 - metric, np.arange(n_docs, dtype=np.uint32), vals2)])
    for nc in (2, 4):
        fields = np.array([0, f1] * (nc // 2), dtype=np.uint32)
        dims = [dim if f == 0 else d2 for f in fields]
        qv = torch.from_numpy(np.concatenate([corpus.unit_vectors(nq, d, seed=40 + i) for i, d in enumerate(dims)],
                                             axis=1)).cuda()
        al = torch.full((nq, nc), 0.5, dtype=torch.float32, device="cuda")
        def run_f():
            N.check(L.slg_rerank_fields_batch_device(ix._h, nq, nc, fields.ctypes.data, qv.data_ptr(), al.data_ptr(), None,
                                                     cd.data_ptr(), cs.data_ptr(), cb.data_ptr(), cc.data_ptr(), ncand, k_out,
                                                     od.data_ptr(), os_.data_ptr(), osc.data_ptr(), ov.data_ptr(), oc.data_ptr()))
        for _ in range(2):
            run_f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            run_f()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 5
        gb = nq * ncand * sum(dims) * 4 / 1e9
        print(f"fields kernel, {nc} clauses over 2 fields ({dim}-d + {d2}-d): {ms:.3f} ms, row bytes {gb:.2f} GB -> {gb / ms:.2f} TB/s", flush=True)
    del vals2
    ix.close()
