#!/usr/bin/env python3
"""bench.py — queries/sec of the batched BM25 top-k scorer on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic queries: the partition,
scoring and merge kernels of `slg_batch_run` with the index and the planned batch already
resident in HBM (plus, for N > 1, the RCCL all-gather of the per-rank top-k).

N = 1 workload = BASELINE.json configs[1]: 1M-doc synthetic Zipf corpus (avg 256 tokens),
3-term OR queries, batch = 1024, top-10 (k = limit + 1 = 11).
N > 1: query batches shard across GPUs — every rank holds a replica of the index, scores its
own 1024-query batch, and the per-rank top-k are exchanged with one all-gather ("weak" scaling:
per-GPU work fixed).  `--mode shard` instead shards the INDEX (configs[3] shape): every rank
scores all queries against its own segment, all-gather, device merge.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c5", "small"])
    ap.add_argument("--dim", type=int, default=768, help="vector dimension (config c5)")
    ap.add_argument("--mode", default="replica", choices=["replica", "shard"])
    ap.add_argument("--docs", type=int, default=0)
    ap.add_argument("--nq", type=int, default=0)
    ap.add_argument("--terms", type=int, default=0)
    ap.add_argument("--limit", type=int, default=0)
    ap.add_argument("--strategy", default="wand", choices=["bm25", "wand", "bmw"])
    ap.add_argument("--inflight", type=int, default=2,
                    help="prepared batches in flight, each on its own HIP stream (steps go "
                         "round-robin over them; every step still does the whole batch's work)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-queries", type=int, default=0, help="queries in the CPU sample")
    ap.add_argument("--check", type=int, default=64, help="queries parity-checked vs the oracle")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) even for one rank, to exercise the all-gather path")
    return ap.parse_args()


CONFIGS = {
    # name: (docs, vocab, corpus_seed, nq, terms, limit)
    "c2": (1_000_000, 1 << 18, 42, 1024, 3, 10),
    "c3": (10_000_000, 1 << 20, 43, 4096, 5, 100),
    "small": (100_000, 1 << 15, 42, 256, 3, 10),
    # config 5: BM25 top-1000 (k = 1001) -> cosine rerank over 768-d f32 vectors -> top-10
    "c5": (1_000_000, 1 << 18, 42, 1024, 3, 1000),
}


class _DevArray:
    """Expose a raw device pointer to torch through __cuda_array_interface__."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr,
                                         "data": (int(ptr), False), "version": 2}


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    from searchlite_amd import corpus, searcher

    n_docs, vocab, cseed, nq, T, limit = CONFIGS[args.config]
    n_docs = args.docs or n_docs
    nq = args.nq or nq
    T = args.terms or T
    limit = args.limit or limit
    k = limit + 1  # api/reader.rs:2615-2619
    strategy = {"bm25": searcher.Bm25, "wand": searcher.Wand, "bmw": searcher.Bmw}[args.strategy]
    host_cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, min(32, host_cores // max(1, world)))

    t0 = time.time()
    shard_mode = args.mode == "shard" and (world > 1 or args.force_dist)
    seg_seed = cseed + (rank if shard_mode else 0)
    seg = corpus.zipf_segment(n_docs, vocab, seed=seg_seed, n_threads=threads)
    q_seed = 7 + (0 if shard_mode else rank)
    offs, terms, w = corpus.zipf_queries(nq, T, seed=q_seed, vocab=vocab)
    t_corpus = time.time() - t0

    rerank = args.config == "c5"
    if rerank:
        seg.vec_dim, seg.vec_metric = args.dim, 0
        seg.vec_offsets = np.arange(n_docs, dtype=np.uint32)
        seg.vec_values = corpus.unit_vectors(n_docs, args.dim, seed=11)
    index = searcher.GpuIndex([seg], device=local_rank)
    stream = torch.cuda.current_stream()
    index.set_stream(stream.cuda_stream)
    # several batches in flight (separate work buffers, separate HIP streams): the partition /
    # merge kernels of one batch overlap the scoring kernel of another.  Not combined with the
    # rerank stage, which runs on the index stream.
    inflight = max(1, args.inflight) if not rerank else 1
    batches = [index.prepare(offs, terms, w, k, strategy) for _ in range(inflight)]
    streams = [torch.cuda.Stream() for _ in range(inflight)] if inflight > 1 else [stream]
    if inflight > 1:
        for b_, s_ in zip(batches, streams):
            b_.set_stream(s_.cuda_stream)
    batch = batches[0]
    info = batch.info()
    d_doc, d_seg, d_score, d_count = batch.device_results()
    t_doc = torch.as_tensor(_DevArray(d_doc, (nq, k), "<i4"), device="cuda")
    t_seg = torch.as_tensor(_DevArray(d_seg, (nq, k), "<i4"), device="cuda")
    t_score = torch.as_tensor(_DevArray(d_score, (nq, k), "<f4"), device="cuda")
    t_count = torch.as_tensor(_DevArray(d_count, (nq,), "<i4"), device="cuda")
    if use_dist:
        t_blocks, g_blocks = [], []
        for b_ in batches:
            blk_ptr, blk_bytes = b_.device_result_block()
            t_blocks.append(torch.as_tensor(_DevArray(blk_ptr, (blk_bytes // 4,), "<i4"), device="cuda"))
            g_blocks.append(torch.empty((world * (blk_bytes // 4),), dtype=torch.int32, device="cuda"))
        t_block, g_block = t_blocks[0], g_blocks[0]
        m_doc = torch.empty((nq, k), dtype=torch.int32, device="cuda")
        m_seg = torch.empty_like(m_doc)
        m_score = torch.empty((nq, k), dtype=torch.float32, device="cuda")
        m_count = torch.empty((nq,), dtype=torch.int32, device="cuda")

    if rerank:
        seg.vec_values = None  # staged in HBM; free the host copy
        k_out = 10
        qv = torch.from_numpy(corpus.unit_vectors(nq, args.dim, seed=12)).cuda()
        alpha = torch.full((nq,), 0.5, dtype=torch.float32, device="cuda")
        r_doc = torch.empty((nq, k_out), dtype=torch.int32, device="cuda")
        r_seg = torch.empty_like(r_doc)
        r_score = torch.empty((nq, k_out), dtype=torch.float32, device="cuda")
        r_vec = torch.empty_like(r_score)
        r_count = torch.empty((nq,), dtype=torch.int32, device="cuda")
        ev_a = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
        ev_b = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    step_no = [0]
    turn = [0]
    gathered = [None] * inflight

    def shard_merge(gblk):  # api/reader.rs:2776-2778 across shards (index stream = default stream)
        n_ = nq * k
        gb = gblk.view(world, -1)
        g_doc = gb[:, :n_].contiguous()
        g_seg = gb[:, n_:2 * n_].contiguous()
        g_score = gb[:, 2 * n_:3 * n_].contiguous()
        g_count = gb[:, 3 * n_:].contiguous()
        index.merge_shards_device(world, nq, k, g_doc.data_ptr(), g_seg.data_ptr(),
                                  g_score.data_ptr(), g_count.data_ptr(), 1,
                                  m_doc.data_ptr(), m_seg.data_ptr(), m_score.data_ptr(),
                                  m_count.data_ptr())

    def step():
        if inflight > 1:
            i = turn[0] % inflight
            turn[0] += 1
            if use_dist and gathered[i] is not None:
                streams[i].wait_event(gathered[i])  # the previous gather of this block is done
            batches[i].run()
            if use_dist:
                # replica mode: one all-gather of this batch's result block.  RCCL stays on the
                # default stream (ordered after the batch's kernels by an event); the batch's
                # stream waits for the gather before the block is overwritten two steps later.
                stream.wait_event(streams[i].record_event())
                dist.all_gather_into_tensor(g_blocks[i], t_blocks[i])
                if shard_mode:
                    shard_merge(g_blocks[i])
                gathered[i] = stream.record_event()
            return
        batch.run()
        if rerank:  # candidates = the BM25 pass's device results (no host round trip)
            timed = step_no[0] >= args.warmup
            if timed:
                ev_a[step_no[0] - args.warmup].record()
            index.rerank_batch_device(nq, qv.data_ptr(), alpha.data_ptr(), d_doc, d_seg, d_score,
                                      d_count, k, k_out, r_doc.data_ptr(), r_seg.data_ptr(),
                                      r_score.data_ptr(), r_vec.data_ptr(), r_count.data_ptr())
            if timed:
                ev_b[step_no[0] - args.warmup].record()
            step_no[0] += 1
        if use_dist:
            # per-rank top-k exchanged over xGMI in ONE all-gather: the contiguous block
            # doc|seg|score|count = (3k+1)*Q*4 bytes per rank
            dist.all_gather_into_tensor(g_block, t_block)
            if shard_mode:
                shard_merge(g_block)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    if inflight == 1:
        index.profile(True)
        index.profile_read()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t1
    if inflight > 1:
        # scoring-kernel duration (HIP events on the kernel's stream): measured on one batch
        # running alone, after the timed region, so overlap with other batches does not blur it
        index.profile(True)
        index.profile_read()
        for _ in range(max(5, min(args.steps, 20))):
            batch.run()
        fence()
    n_launch, kern_ms = index.profile_read()
    index.profile(False)
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        pp = torch.tensor([float(info["n_postings"])], dtype=torch.float64, device="cuda")
        dist.all_reduce(pp, op=dist.ReduceOp.SUM)
        total_postings = float(pp.item())
    else:
        total_postings = float(info["n_postings"])

    ms_per_step = elapsed / args.steps * 1e3
    queries_per_step = nq if shard_mode else nq * world
    value = queries_per_step / (elapsed / args.steps)

    out = None
    if rank == 0:
        kern_avg_ms = kern_ms / max(n_launch, 1)
        alg_bytes = info["algorithmic_bytes"]  # 12 B/posting + 8*k*nq
        achieved = alg_bytes / (kern_avg_ms * 1e-3) / 1e9 if kern_avg_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_c2.json")
        if args.config == "c2" and world == 1 and os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "queries/sec at top-10 (batch=1024) + achieved HBM GB/s vs peak",
            "value": round(value, 1), "unit": "queries/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong" if shard_mode else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{n_docs} synthetic Zipf docs (avg 256 tokens, V={vocab}, "
                                   f"s=1.0, seed {cseed}), {T}-term OR, batch={nq} per GPU, "
                                   f"top-{limit} (k={k}), strategy={args.strategy}, "
                                   f"{'index-sharded' if shard_mode else 'query-sharded replicas'}",
                       "postings_per_batch": int(info["n_postings"]),
                       "slices": int(info["n_slices"]), "batches_in_flight": inflight,
                       "all_ranks_postings": int(total_postings),
                       "corpus_build_s": round(t_corpus, 1)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "kernel": "score_uniform_kernel" if T <= 4 else "score_multi_kernel",
                         "kernel_ms": round(kern_avg_ms, 4), "launches": n_launch,
                         "algorithmic_bytes_per_launch": int(alg_bytes)},
        }

    if rank == 0 and rerank:
        rr_ms = sum(a.elapsed_time(b_) for a, b_ in zip(ev_a, ev_b)) / args.steps
        cands = int(t_count.sum().item())
        rr_bytes = 4 * args.dim * cands + 8 * cands + 4 * args.dim * nq
        out["rerank"] = {"kernel": "rerank_kernel", "kernel_ms": round(rr_ms, 4),
                         "candidates": cands, "algorithmic_bytes": rr_bytes,
                         "achieved_GBps": round(rr_bytes / (rr_ms * 1e-3) / 1e9, 1),
                         "frac_of_hbm_peak": round(rr_bytes / (rr_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "note": "BM25 top-1000 candidates reranked by cosine (dot of unit vectors) "
                                 "with alpha=0.5 blend to top-10; value = whole pipeline"}
    # ---- parity spot-check + CPU baseline (rank 0, N = 1 only for the baseline) ----
    if rank == 0:
        from oracle import oracle as O
        got = batch.fetch()
        for b_ in batches[1:]:  # every in-flight batch computed the same results
            other = b_.fetch()
            if not all(np.array_equal(np.asarray(x).view(np.uint32), np.asarray(y).view(np.uint32))
                       for x, y in zip(got[:4], other[:4])):
                raise SystemExit("bench.py: in-flight batches disagree")
        nchk = min(args.check, nq)
        if nchk:
            want = O.search_batch([seg], offs[:nchk + 1], terms[:nchk * T], w[:nchk * T], k,
                                  strategy=O.BM25, n_threads=threads)
            ok = True
            for q in range(nchk):
                n = int(want[3][q])
                ok &= int(got[3][q]) == n and np.array_equal(got[0][q, :n], want[0][q, :n]) \
                    and np.array_equal(got[2][q, :n].view(np.uint32), want[2][q, :n].view(np.uint32))
            out["parity"] = {"queries_checked": nchk, "bit_exact": bool(ok)}
            if not ok:
                print(json.dumps(out))
                raise SystemExit("bench.py: GPU results differ from the oracle")
        if rerank and nchk:
            # rerank spot check against the oracle (tolerance 1e-5 on blended scores)
            hv = index.segments[0]
            vv = corpus.unit_vectors(n_docs, args.dim, seed=11)
            qh = qv.cpu().numpy()
            worst = 0.0
            for q in range(min(nchk, 8)):
                n = int(got[3][q])
                wd, ws, _ = O.rerank(0, hv.vec_offsets, vv, qh[q], 0.5, got[0][q, :n], got[2][q, :n], 10)
                gs = r_score[q].cpu().numpy()
                worst = max(worst, float(np.abs(gs[:len(ws)] - ws).max()))
            out["rerank"]["max_abs_err_vs_oracle"] = worst
            del vv
        if world == 1 and not args.no_cpu_baseline:
            cores = host_cores
            ncpu = min(args.cpu_queries or nq, nq)
            co, ct, cw = offs[:ncpu + 1], terms[:ncpu * T], w[:ncpu * T]
            ostrat = {"bm25": O.BM25, "wand": O.WAND, "bmw": O.BMW}[args.strategy]
            # strict baseline: scorer only, min_doc_len cached (a cache the reference lacks)
            reps, t_cpu = 0, 0.0
            while t_cpu < 10.0 and reps < 50:
                tc = time.perf_counter()
                O.search_batch([seg], co, ct, cw, k, strategy=ostrat, n_threads=cores,
                               cache_min_len=True)
                t_cpu += time.perf_counter() - tc
                reps += 1
            strict = ncpu * reps / t_cpu
            # faithful: TermState::new rescans all doc lengths per term per query (wand.rs:111-125)
            nf = min(ncpu, 256)
            tc = time.perf_counter()
            O.search_batch([seg], offs[:nf + 1], terms[:nf * T], w[:nf * T], k, strategy=ostrat,
                           n_threads=cores, cache_min_len=False)
            faithful = nf / (time.perf_counter() - tc)
            out["cpu_baseline"] = {
                "value": round(strict, 1), "unit": "queries/s", "cores": cores, "kind": "port",
                "sample": f"{ncpu} queries of the same batch x {reps} reps, oracle "
                          f"{args.strategy} (C restatement of searchlite-core's scorer, "
                          f"pre-decoded postings, cached doc lengths and min_doc_len), "
                          f"{cores} threads, one query per thread",
                "faithful_value": round(faithful, 1),
                "faithful_note": "same, but with the reference's per-term O(N) min_doc_len scan "
                                 f"(wand.rs:111-125) on {nf} queries",
                "gpu_over_cpu": round(value / strict, 1)}
        print(json.dumps(out), flush=True)

    for b_ in batches:
        b_.close()
    index.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
