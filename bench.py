#!/usr/bin/env python3
"""bench.py — queries/sec of the batched BM25 top-k scorer on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic queries.

`value` (N = 1, BASELINE.json configs[1]: 1M-doc Zipf corpus, 3-term OR, batch 1024, top-10) is
the SURVEY section 8(d) figure: the index is resident in HBM, every step takes a FRESH query
batch as host arrays through the C ABI — slg_batch_prepare (plan + H2D) -> slg_batch_run (three
kernels) -> slg_batch_fetch (D2H into host arrays) -> destroy — from `--host-threads` caller
threads (the reference serves one request per thread, searchlite-http/src/lib.rs:640-643), each
keeping two batches going on two HIP streams of its own (it launches the next batch before it
collects the previous one).  The batches rotate over `--rotate` distinct query sets, so
the posting working set (8 x 262 MB for config 2) is far beyond the 256 MiB Infinity Cache.
Reported beside it (config.*): the device-resident rate of pre-planned batches, and the scoring
kernel's own duration (HIP events on its stream, batches rotating, one at a time) for `roofline`.

N > 1 (launched by the driver through torch.distributed.run, or by `--gpus N` itself, which starts
the N ranks as a child torchrun before touching any GPU): query batches shard across replicas of
the index (weak scaling, no data-path collective: every rank serves its own queries).  In addition the BASELINE configs[3] shape is timed
(`c4` object; `--config c4` makes it the main workload): the 10M-doc corpus as 8 segments of
1.25M docs (seeds 43..50) index-sharded over the N ranks, batch 8192 x 5 terms, top-100; every
rank scores all queries against its segments, ONE ncclAllGather of the (3k+1)*Q*4-byte result
blocks over xGMI behind the C ABI (slg_batch_run_sharded), device merge (strong scaling: the same 8
segments at every N).  torch.distributed carries only the control plane (rendezvous, the 128-byte
communicator id, barriers and the max-over-ranks of the timings).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import itertools
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

CONFIGS = {
    # name: (docs, vocab, corpus_seed, nq, terms, limit)
    "c2": (1_000_000, 1 << 18, 42, 1024, 3, 10),
    "c3": (10_000_000, 1 << 20, 43, 4096, 5, 100),
    "small": (100_000, 1 << 15, 42, 256, 3, 10),
    # config 4: 8 segments x 1.25M docs (seeds 43 + s), index-sharded over the ranks
    "c4": (1_250_000, 1 << 20, 43, 8192, 5, 100),
    # config 5: BM25 top-1000 (k = 1001) -> cosine rerank over 768-d f32 vectors -> top-10
    "c5": (1_000_000, 1 << 18, 42, 1024, 3, 1000),
    # multi-field query strings (the reference's default `fields: None` = all text fields,
    # api/reader.rs:2576-2586): config 2's corpus split over 4 fields (4 x ~64 tokens per doc, seeds
    # 42..45), 2-word query strings = 8 scored lists in 2 ScorePlan leaves (Sum of leaves,
    # query/planner.rs:354-360).  `terms` = words per query
    "mf": (1_000_000, 1 << 18, 42, 1024, 2, 10),
}
MF_FIELDS = 4
C4_SEGMENTS = 8


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--dim", type=int, default=768, help="vector dimension (config c5)")
    ap.add_argument("--docs", type=int, default=0)
    ap.add_argument("--nq", type=int, default=0)
    ap.add_argument("--terms", type=int, default=0)
    ap.add_argument("--limit", type=int, default=0)
    ap.add_argument("--strategy", default="wand", choices=["bm25", "wand", "bmw"])
    ap.add_argument("--regions", type=int, default=9,
                    help="host-inclusive leg: the timed --steps region is repeated this many times; value = median")
    ap.add_argument("--rotate", type=int, default=8, help="distinct query sets the steps rotate over")
    ap.add_argument("--host-threads", type=int, default=8,
                    help="caller threads of the host-inclusive leg (each: prepare -> run -> fetch)")
    ap.add_argument("--inflight", type=int, default=2,
                    help="device-resident leg: prepared batches in flight, each on its own HIP stream")
    ap.add_argument("--kernel-leg-only", action="store_true",
                    help="profiling runs: only the rotating, one-at-a-time kernel leg (value = its rate)")
    ap.add_argument("--no-c4", action="store_true", help="N > 1: skip the config-4 (index-sharded) leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-queries", type=int, default=0, help="queries in the CPU sample")
    ap.add_argument("--check", type=int, default=64, help="queries parity-checked vs the oracle")
    ap.add_argument("--c4-threads", type=int, default=4, help="caller threads per rank of the index-sharded leg")
    ap.add_argument("--coalesce-threads", type=int, default=256,
                    help="caller threads of the request-coalescer leg (config c2, N = 1; 0: skip the leg)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) even for one rank")
    return ap.parse_args()


def spawn_ranks_if_needed(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child torchrun and
    exit with its code.  Runs before anything in this process touches a GPU."""
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is not None:
        if int(world_env) != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_env}")
        return
    if args.gpus <= 1:
        return
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


class _DevArray:
    """Expose a raw device pointer to torch through __cuda_array_interface__."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr,
                                         "data": (int(ptr), False), "version": 2}


def main():
    args = parse_args()
    spawn_ranks_if_needed(args)
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    from searchlite_amd import corpus, searcher

    host_cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    visible_cpus = host_cores
    cpu_quota = None
    try:  # a cgroup quota below the visible CPU count is what the process really has (GPU boxes of this pool:
        # 256 CPUs visible, cpu.max = 16 CPUs): the CPU baseline runs on that many threads and says so
        q_, p_ = open("/sys/fs/cgroup/cpu.max").read().split()
        if q_ != "max":
            cpu_quota = int(q_) / int(p_)
            host_cores = max(1, min(host_cores, int(cpu_quota + 0.5)))
    except Exception:  # noqa: BLE001
        pass
    gen_threads = max(1, min(32, host_cores // max(1, world)))
    strategy = {"bm25": searcher.Bm25, "wand": searcher.Wand, "bmw": searcher.Bmw}[args.strategy]
    stream = torch.cuda.current_stream()

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if not use_dist:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(x):
        if not use_dist:
            return x
        t = torch.tensor([float(x)], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    def query_sets(nq, T, vocab, n_segs, first_seed, n_sets):
        out = []
        for j in range(n_sets):
            offs, terms, w = corpus.zipf_queries(nq, T, seed=first_seed + j, vocab=vocab)
            # term id == rank - 1 in every zipf segment: the same id in each segment's dictionary
            t2 = np.ascontiguousarray(np.repeat(terms.reshape(-1, 1), n_segs, axis=1))
            out.append((offs, t2, w))
        return out

    # ------------------------------------------------------------------------------------------
    # index-sharded workload (config 4 shape): device-resident index, fresh host queries per step,
    # ONE all-gather of the per-rank result blocks + device merge, merged top-k back on the host
    # ------------------------------------------------------------------------------------------
    def run_sharded(cfg_name, steps, warmup):
        n_docs, vocab, cseed, nq, T, limit = CONFIGS[cfg_name]
        n_docs = args.docs or n_docs
        nq = (args.nq or nq) if args.config == cfg_name else nq
        k = limit + 1
        if C4_SEGMENTS % world:
            raise SystemExit(f"config c4 needs a rank count that divides {C4_SEGMENTS}")
        per_rank = C4_SEGMENTS // world
        t0 = time.time()
        segs = [corpus.zipf_segment(n_docs, vocab, seed=cseed + rank * per_rank + s, n_threads=gen_threads)
                for s in range(per_rank)]
        t_corpus = time.time() - t0
        index = searcher.GpuIndex(segs, device=local_rank)
        index.set_stream(stream.cuda_stream)
        del segs
        qs = query_sets(nq, T, vocab, per_rank, 7, max(2, args.rotate))
        n_flight = 2
        streams = [torch.cuda.Stream() for _ in range(n_flight)]
        # the shard group: RCCL behind the C ABI (slg_shard_group_create = ncclCommInitRank).  The
        # 128-byte communicator id travels over the control plane once; the data path below has no
        # torch.distributed call: slg_batch_run_sharded = kernels + ONE ncclAllGather + device merge
        if use_dist:
            box = [searcher.shard_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            uid = box[0]
        else:
            uid = searcher.shard_unique_id()
        group = searcher.ShardGroup(index, rank, world, uid, per_rank)
        # The caller threads are native (lib/libslg_harness.so), as in the replica workload: `c4_threads`
        # threads per rank, each keeping two batches going on two HIP streams of its own — prepare
        # (host planning + H2D of the descriptors) of one batch overlaps the kernels, the all-gather
        # and the merge of the others.  With several threads the ranks' collectives are ordered by the
        # step number (slg_batch_run_sharded_seq): the same on every rank.
        import ctypes as C
        from searchlite_amd import build as sbuild
        L = C.CDLL(sbuild.build_harness())
        L.slh_create.restype = C.c_void_p
        L.slh_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_uint32, C.c_uint32, C.c_int]
        L.slh_set_group.argtypes = [C.c_void_p, C.c_void_p]
        L.slh_run.restype = C.c_int
        L.slh_run.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
        L.slh_error.restype = C.c_char_p
        L.slh_error.argtypes = [C.c_void_p]
        L.slh_first_result.restype = C.c_int
        L.slh_first_result.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.slh_destroy.argtypes = [C.c_void_p]
        L.slh_stats.argtypes = [C.c_void_p, C.c_void_p]
        L.slh_reset_stats.argtypes = [C.c_void_p]
        keep = [(np.ascontiguousarray(o, np.uint32), np.ascontiguousarray(t, np.uint32),
                 np.ascontiguousarray(w_, np.float32)) for o, t, w_ in qs]
        ptrs = lambda j: (C.c_void_p * len(keep))(*[x[j].ctypes.data for x in keep])
        c4_threads = max(1, args.c4_threads)
        hh = L.slh_create(index._h, local_rank, c4_threads, len(keep), ptrs(0), ptrs(1), ptrs(2), nq, k, strategy)
        L.slh_set_group(hh, group._h)
        postings = []
        for j in range(len(qs)):  # planning facts of every query set (not timed)
            bb = index.prepare(*qs[j], k, strategy)
            postings.append(bb.info())
            bb.close()

        def run_steps(n, first):
            if L.slh_run(hh, first, n) != 0:
                raise RuntimeError("host harness (sharded): " + L.slh_error(hh).decode())

        warm = max(warmup, 2 * c4_threads)
        run_steps(warm, 0)
        L.slh_reset_stats(hh)
        index.profile(True)
        group.stats()
        fence()
        t1 = time.perf_counter()
        run_steps(steps, warm)
        fence()
        elapsed = max_over_ranks(time.perf_counter() - t1)
        index.profile(False)
        hs = (C.c_double * 4)()
        L.slh_stats(hh, hs)
        gstats = group.stats()
        d0 = np.zeros((nq, k), np.uint32)
        s0 = np.zeros((nq, k), np.uint32)
        sc0 = np.zeros((nq, k), np.float32)
        c0 = np.zeros(nq, np.uint32)
        if not L.slh_first_result(hh, 0, d0.ctypes.data, s0.ctypes.data, sc0.ctypes.data, c0.ctypes.data):
            raise RuntimeError("host harness (sharded): no result for query set 0")
        last_res = [(d0, s0, sc0, c0)]
        L.slh_destroy(hh)
        res = tuple(np.asarray(x).copy() for x in last_res[0])
        last_q = qs[0]  # (the harness keeps the first result of every query set: set 0)
        info = index.info()
        # the scoring kernel alone (HIP events on its launch stream, one launch at a time, query sets rotating)
        index.profile_read()
        index.profile(True)
        iso = [index.prepare(*qs[j], k, strategy) for j in range(min(2, len(qs)))]
        for _ in range(2):
            for bb in iso:
                bb.run()
        fence()
        n_launch, kern_ms = index.profile_read()
        index.profile(False)
        iso_postings = float(np.mean([bb.info()["n_postings"] for bb in iso]))
        for bb in iso:
            bb.close()
        kern_avg = kern_ms / max(n_launch, 1)
        alg = 12.0 * iso_postings + 8.0 * k * nq * per_rank  # per posting, + a top-k row per sub-query
        roof = {"bound": "hbm", "achieved": round(alg / (kern_avg * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(alg / (kern_avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                "kernel": "score_uniform4_kernel", "kernel_ms": round(kern_avg, 4), "launches": n_launch,
                "algorithmic_bytes_per_launch": int(alg),
                "bytes_note": "this rank's shard: 12 B x postings scored + 8*k per sub-query (SURVEY 8d)"}
        group.close()
        index.close()
        out = {"workload": f"{C4_SEGMENTS} segments x {n_docs} synthetic Zipf docs (V={vocab}, seeds {cseed}.."
                           f"{cseed + C4_SEGMENTS - 1}) index-sharded over {world} GPU(s), {T}-term OR, "
                           f"batch={nq}, top-{limit} (k={k}), strategy={args.strategy}; fresh host query "
                           f"batch per step, slg_batch_run_sharded (ONE ncclAllGather of (3k+1)*Q*4 B per rank "
                           f"behind the C ABI + device merge), merged top-k copied to the host",
               "queries_per_s": round(nq / (elapsed / steps), 1), "ms_per_step": round(elapsed / steps * 1e3, 4),
               "steps": steps, "scaling": "strong", "segments_per_rank": per_rank,
               "caller_threads_per_rank": c4_threads,
               # rank 0's view of one step (means over the timed region): host time inside the C ABI per batch
               # (a thread keeps two batches going: plan_ms overlaps the device work of other batches) and the
               # device time of the batch's three phases (gather_ms includes waiting for the slowest rank)
               "per_rank": {"plan_ms": round(hs[0], 3), "launch_ms": round(hs[1], 3), "fetch_wait_ms": round(hs[2], 3),
                            "kernel_ms": round(gstats["kernel_ms"], 3), "gather_ms": round(gstats["gather_ms"], 3),
                            "merge_ms": round(gstats["merge_ms"], 3), "runs_timed": gstats["runs"]},
               "postings_per_batch_this_rank": int(np.mean([p["n_postings"] for p in postings])),
               "index_postings_this_rank": int(info["n_postings"]), "corpus_build_s": round(t_corpus, 1)}
        out["roofline_this_rank"] = roof
        return out, res, last_q, (n_docs, vocab, cseed, nq, T, k)

    n_docs, vocab, cseed, nq, T, limit = CONFIGS[args.config]
    n_docs = args.docs or n_docs
    nq = args.nq or nq
    T = args.terms or T
    limit = args.limit or limit
    k = limit + 1  # api/reader.rs:2615-2619
    out = None

    if args.config == "c4":
        c4, res, last_q, _ = run_sharded("c4", args.steps, args.warmup)
        if rank == 0:
            out = {"metric": "queries/sec at top-10 (batch=1024) + achieved HBM GB/s vs peak",
                   "value": c4["queries_per_s"], "unit": "queries/s", "n_gpus": world, "steps": args.steps,
                   "warmup": args.warmup, "ms_per_step": c4["ms_per_step"], "higher_is_better": True,
                   "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                   "config": c4, "roofline": c4.pop("roofline_this_rank")}
            if world == 1 and args.check:
                from oracle import oracle as O
                segs = [corpus.zipf_segment(n_docs, vocab, seed=cseed + s, n_threads=gen_threads)
                        for s in range(C4_SEGMENTS)]
                nchk = min(args.check, 16)
                offs, terms, w = last_q
                want = O.search_batch(segs, offs[:nchk + 1], terms[:nchk * T], w[:nchk * T], k,
                                      strategy=O.BM25, n_threads=gen_threads)
                ok = all(int(res[3][q]) == int(want[3][q]) and
                         np.array_equal(res[0][q].view(np.uint32), want[0][q]) and
                         np.array_equal(res[1][q].view(np.uint32), want[1][q]) and
                         np.array_equal(res[2][q].view(np.uint32), want[2][q].view(np.uint32))
                         for q in range(nchk))
                out["parity"] = {"queries_checked": nchk, "bit_exact": bool(ok)}
                if not ok:
                    print(json.dumps(out))
                    raise SystemExit("bench.py: GPU results differ from the oracle")
            print(json.dumps(out), flush=True)
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ------------------------------------------------------------------------------------------
    # replica workload (configs 2 / 3 / 5 / small)
    # ------------------------------------------------------------------------------------------
    t0 = time.time()
    plans = args.config == "mf"  # score plans: every query carries q_leaf (a leaf per word)
    seg = corpus.zipf_multifield_segment(n_docs, vocab, MF_FIELDS, seed=cseed, n_threads=gen_threads) if plans \
        else corpus.zipf_segment(n_docs, vocab, seed=cseed, n_threads=gen_threads)
    t_corpus = time.time() - t0
    rerank = args.config == "c5"
    if rerank:
        seg.vec_dim, seg.vec_metric = args.dim, 0
        seg.vec_offsets = np.arange(n_docs, dtype=np.uint32)
        seg.vec_values = corpus.unit_vectors(n_docs, args.dim, seed=11)
    index = searcher.GpuIndex([seg], device=local_rank)
    index.set_stream(stream.cuda_stream)
    n_sets = max(1, args.rotate)
    qs = query_sets(nq, T, vocab, 1, 7 + rank * n_sets, n_sets)  # seeds 7.. (rank 0), distinct per rank
    q_leaves = None
    if plans:
        mq = [corpus.multifield_queries(nq, T, MF_FIELDS, vocab, seed=7 + rank * n_sets + j) for j in range(n_sets)]
        qs = [(o, t.reshape(-1, 1), w_) for o, t, w_, _ in mq]
        q_leaves = [l for _, _, _, l in mq]

    # ---- leg 1 (`value`): host arrays in -> prepare -> run -> fetch -> host arrays out ----
    n_thr = max(1, args.host_threads)
    first_results = {}

    class HostPool:
        """The caller threads (a server's request threads): lib/libslg_harness.so — the same loop a
        Rust / C++ host of the C ABI runs (searchlite serves one request per OS thread,
        searchlite-http/src/lib.rs:640-643), without Python's interpreter lock in the measurement.
        Persistent threads, created and warmed before the timed region, fed step numbers round-robin;
        a thread keeps TWO batches going on two HIP streams of its own (prepare + run of the next
        batch, then fetch + destroy of the previous one), so one batch's host round trip never
        leaves the GPU without queued work.  run(n) returns when n steps have been fetched."""

        def __init__(self):
            import ctypes as C
            from searchlite_amd import build as sbuild
            L = C.CDLL(sbuild.build_harness())
            L.slh_create.restype = C.c_void_p
            L.slh_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_uint32, C.c_uint32, C.c_int]
            L.slh_run.restype = C.c_int
            L.slh_run.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
            L.slh_error.restype = C.c_char_p
            L.slh_error.argtypes = [C.c_void_p]
            L.slh_first_result.restype = C.c_int
            L.slh_first_result.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
            L.slh_destroy.argtypes = [C.c_void_p]
            L.slh_stats.argtypes = [C.c_void_p, C.c_void_p]
            L.slh_reset_stats.argtypes = [C.c_void_p]
            self.L, self.C = L, C
            self.keep = [(np.ascontiguousarray(o, np.uint32), np.ascontiguousarray(t, np.uint32),
                          np.ascontiguousarray(w_, np.float32)) for o, t, w_ in qs]
            ptrs = lambda j: (C.c_void_p * n_sets)(*[x[j].ctypes.data for x in self.keep])
            self.h = L.slh_create(index._h, local_rank, n_thr, n_sets, ptrs(0), ptrs(1), ptrs(2), nq, k, strategy)

        def run(self, n_steps, first):
            if self.L.slh_run(self.h, first, n_steps) != 0:
                raise RuntimeError("host harness: " + self.L.slh_error(self.h).decode())

        def first_result(self, j):
            d = np.zeros((nq, k), np.uint32)
            s_ = np.zeros((nq, k), np.uint32)
            sc = np.zeros((nq, k), np.float32)
            c = np.zeros(nq, np.uint32)
            ok = self.L.slh_first_result(self.h, j, d.ctypes.data, s_.ctypes.data, sc.ctypes.data, c.ctypes.data)
            return (d, s_, sc, c) if ok else None

        def stats(self):
            import ctypes as C
            out = (C.c_double * 4)()
            self.L.slh_stats(self.h, out)
            return {"prepare_ms": round(out[0], 4), "run_ms": round(out[1], 4), "fetch_ms": round(out[2], 4),
                    "batches": int(out[3]),
                    "is": "mean time per batch a caller thread spends inside slg_batch_prepare / set_stream + run / "
                          "fetch + destroy, over the timed regions (warm-up excluded); a thread keeps two batches "
                          "going, so fetch mostly waits for the batch's kernels behind the other threads' batches"}

        def close(self):
            self.L.slh_destroy(self.h)
            self.h = None

    value = ms_per_step = value_spread = host_ms = None
    if not rerank and not plans and not args.kernel_leg_only:
        pool = HostPool()
        # untimed: --warmup steps, and at least three batches per caller thread so that every
        # thread has run and the library's buffer pool holds a set of work buffers per batch in flight
        host_warm = max(args.warmup, 3 * n_thr)
        pool.run(host_warm, 0)
        pool.L.slh_reset_stats(pool.h)
        # the timed region = EXACTLY --steps steps between two fences; it is repeated --regions times
        # (a 20-step region is 2.5 ms: 2.5 steps per caller thread with the pipeline's fill and drain
        # inside) and `value` is the MEDIAN region; the spread is reported beside it
        region_s = []
        for r in range(max(1, args.regions)):
            fence()
            t1 = time.perf_counter()
            pool.run(args.steps, host_warm + r * args.steps)
            fence()
            region_s.append(max_over_ranks(time.perf_counter() - t1))
        hp0 = pool.first_result(0)
        if hp0 is not None:
            first_results[0] = hp0
        host_ms = pool.stats()
        pool.close()
        elapsed = float(np.median(region_s))
        ms_per_step = elapsed / args.steps * 1e3
        value = nq * world / (elapsed / args.steps)
        value_spread = {"regions": len(region_s), "steps_per_region": args.steps,
                        "min": round(nq * world / (max(region_s) / args.steps), 1),
                        "max": round(nq * world / (min(region_s) / args.steps), 1),
                        "value_is": "median over the regions"}

    # ---- leg 2: device-resident pre-planned batches, rotating, `inflight` streams ----
    inflight = max(1, args.inflight)
    batches = [index.prepare(*q, k, strategy, q_leaf=None if q_leaves is None else q_leaves[j])
               for j, q in enumerate(qs)]
    infos = [b.info() for b in batches]
    streams = [torch.cuda.Stream() for _ in range(inflight)] if inflight > 1 else [stream]
    if inflight > 1:
        for j, b in enumerate(batches):
            b.set_stream(streams[j % inflight].cuda_stream)
    if rerank:
        seg.vec_values = None  # staged in HBM; free the host copy
        k_out = 10
        d_res = [b.device_results() for b in batches]
        cnt_t = [torch.as_tensor(_DevArray(d[3], (nq,), "<i4"), device="cuda") for d in d_res]
        qv = torch.from_numpy(corpus.unit_vectors(nq, args.dim, seed=12)).cuda()
        alpha = torch.full((nq,), 0.5, dtype=torch.float32, device="cuda")
        # one set of rerank outputs per pipeline in flight (r_* = set 0: the parity check reads it)
        r_sets = [(torch.empty((nq, k_out), dtype=torch.int32, device="cuda"),
                   torch.empty((nq, k_out), dtype=torch.int32, device="cuda"),
                   torch.empty((nq, k_out), dtype=torch.float32, device="cuda"),
                   torch.empty((nq, k_out), dtype=torch.float32, device="cuda"),
                   torch.empty((nq,), dtype=torch.int32, device="cuda")) for _ in range(inflight)]
        r_doc, r_seg, r_score, r_vec, r_count = r_sets[0]
        ev_a = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
        ev_b = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    turn = [0]

    def resident_step(timed_idx=None):
        j = turn[0] % n_sets
        turn[0] += 1
        b = batches[j]
        b.run()
        if rerank:  # candidates = the BM25 pass's device results, reranked on the batch's own stream
            st_j = streams[j % inflight] if inflight > 1 else stream
            if timed_idx is not None:
                ev_a[timed_idx].record(st_j)
            rs = r_sets[j % inflight]
            b.rerank_device(1, qv.data_ptr(), alpha.data_ptr(), None, k_out, rs[0].data_ptr(), rs[1].data_ptr(),
                            rs[2].data_ptr(), rs[3].data_ptr(), rs[4].data_ptr())
            if timed_idx is not None:
                ev_b[timed_idx].record(st_j)

    for _ in range(args.warmup):
        resident_step()
    fence()
    t1 = time.perf_counter()
    for i in range(1 if args.kernel_leg_only else args.steps):
        resident_step(i)
    fence()
    res_elapsed = max_over_ranks(time.perf_counter() - t1) * (args.steps if args.kernel_leg_only else 1)
    resident_qps = nq * world / (res_elapsed / args.steps)
    if rerank or plans:
        value, ms_per_step = resident_qps, res_elapsed / args.steps * 1e3

    # ---- leg 3: the scoring kernel alone (HIP events on its launch stream), batches rotating ----
    for b in batches:
        b.set_stream(None)  # back on the index stream: one kernel at a time
    index.profile(True)
    index.profile_read()
    reps = max(2, min(8, (args.steps + n_sets - 1) // n_sets))
    for _ in range(reps):
        for b in batches:
            b.run()
    fence()
    n_launch, kern_ms = index.profile_read()
    index.profile(False)
    # pruned batches: postings in blocks that block skipping never loaded are not credited
    skips = [b.skip_counts() for b in batches]
    if args.kernel_leg_only:
        value = nq * world / (kern_ms / max(n_launch, 1) * 1e-3)
        ms_per_step = kern_ms / max(n_launch, 1)
    total_postings = sum_over_ranks(float(np.mean([i["n_postings"] for i in infos])))
    touched = np.unique(np.concatenate([np.asarray(q[1]).reshape(-1) for q in qs]))
    working_set_postings = int((seg.term_offsets[touched.astype(np.int64) + 1] - seg.term_offsets[touched.astype(np.int64)]).sum())

    if rank == 0:
        kern_avg_ms = kern_ms / max(n_launch, 1)
        exhaustive_bytes = float(np.mean([i["algorithmic_bytes"] for i in infos]))  # 12 B/posting + 8*k*nq
        skipped = float(np.mean([s_[1] for s_ in skips]))
        nonessential = float(np.mean([s_[0] for s_ in skips]))
        alg_bytes = exhaustive_bytes - 12.0 * skipped  # the blocks actually loaded and scored
        achieved = alg_bytes / (kern_avg_ms * 1e-3) / 1e9 if kern_avg_ms > 0 else 0.0
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", f"traffic_{args.config}.json")
        if world == 1 and os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic, traffic_src = tj.get("hbm_bytes_per_launch"), tj.get("source")
            except Exception:  # noqa: BLE001
                traffic = None
        out = {
            "metric": "queries/sec at top-10 (batch=1024) + achieved HBM GB/s vs peak",
            "value": round(value, 1), "unit": "queries/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "value_spread": value_spread,
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"{n_docs} synthetic Zipf docs (avg 256 tokens, V={vocab}, "
                                    f"s=1.0, seed {cseed}), {T}-term OR, batch={nq} per GPU, "
                                    f"top-{limit} (k={k}), strategy={args.strategy}, "
                                    f"query-sharded replicas") if not plans else
                                   (f"{n_docs} synthetic Zipf docs x {MF_FIELDS} text fields (avg 64 tokens each, V={vocab} "
                                    f"per field, s=1.0, seeds {cseed}..{cseed + MF_FIELDS - 1}), {T}-word query strings "
                                    f"over all fields = {T * MF_FIELDS} scored lists in {T} ScorePlan leaves (Sum of "
                                    f"leaves, planner.rs:354-360), batch={nq} per GPU, top-{limit} (k={k}), "
                                    f"strategy={args.strategy}"),
                       "value_is": ("--kernel-leg-only: rate of the scoring kernel alone (profiling run)"
                                    if args.kernel_leg_only else
                                    "device-resident index; every step a fresh host query batch through the "
                                    "C ABI: slg_batch_prepare (plan + H2D) -> run -> fetch (D2H) -> destroy"
                                    if not (rerank or plans) else
                                    "device-resident pre-planned batches with score plans, rotating" if plans else
                                    "device-resident pipeline BM25 top-1000 -> rerank -> top-10 of pre-planned batches"),
                       "host_threads": None if (rerank or plans) else n_thr,
                       "host_call_ms": host_ms,
                       "host_warmup_steps": None if (rerank or plans or args.kernel_leg_only) else host_warm,
                       "rotating_query_sets": n_sets,
                       # 8 B x postings of the DISTINCT lists the rotating query sets touch (the resident
                       # doc-id + impact streams a pass over all sets reads at least once)
                       "posting_working_set_bytes": int(8 * working_set_postings),
                       "kernel_only_qps": round(resident_qps, 1),
                       "kernel_only_ms_per_step": round(res_elapsed / args.steps * 1e3, 4),
                       "kernel_only_is": f"pre-planned device-resident batches, {inflight} in flight",
                       "postings_per_batch": int(np.mean([i["n_postings"] for i in infos])),
                       "slices": int(np.mean([i["n_slices"] for i in infos])),
                       "all_ranks_postings": int(total_postings),
                       "corpus_build_s": round(t_corpus, 1)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_measured_in_run": False,
                         "kernel": (("score_uniform4_kernel<.., PLAN>" if os.environ.get("SLG_NO_UNIFORM_PLANS", "0") == "0"
                                     else "score_multi_kernel<.., 2>") if plans else
                                    {"2": "score_uniform_kernel", "3": "score_uniform3_kernel"}.get(
                                        os.environ.get("SLG_UNIFORM_KERNEL", "4"), "score_uniform4_kernel")
                                    if T <= int(os.environ.get("SLG_UNIFORM_MAX_TERMS", "8")) else "score_multi_kernel"),
                         "kernel_ms": round(kern_avg_ms, 4), "launches": n_launch,
                         "algorithmic_bytes_per_launch": int(alg_bytes),
                         "exhaustive_bytes_per_launch": int(exhaustive_bytes),
                         "postings_in_pruning_classified_lists": int(nonessential),
                         "postings_skipped": int(skipped),
                         "bytes_note": "12 B x postings of every 64-posting block that was loaded and scored + 8*k*Q "
                                       "(SURVEY 8d); blocks that block skipping never loaded are not credited; "
                                       "the exhaustive figure 12 B x all postings is beside it"},
        }
        if rerank:
            rr_pipe_ms = sum(a.elapsed_time(b_) for a, b_ in zip(ev_a, ev_b)) / args.steps
            # the rerank kernel ALONE (one launch at a time on one stream, nothing else on the device): inside the
            # pipeline two batches are in flight and its events also span what the other stream's kernels take
            st0 = streams[0] if inflight > 1 else stream
            n_iso = max(4, min(16, args.steps))
            iso_a = [torch.cuda.Event(enable_timing=True) for _ in range(n_iso)]
            iso_b = [torch.cuda.Event(enable_timing=True) for _ in range(n_iso)]
            for bj in batches:
                bj.set_stream(st0.cuda_stream)  # (leg 3 put the batches back on the index stream)
            fence()
            for i in range(n_iso):
                bj = batches[i % n_sets]
                iso_a[i].record(st0)
                bj.rerank_device(1, qv.data_ptr(), alpha.data_ptr(), None, k_out, r_doc.data_ptr(), r_seg.data_ptr(),
                                 r_score.data_ptr(), r_vec.data_ptr(), r_count.data_ptr())
                iso_b[i].record(st0)
            fence()
            rr_ms = sum(a.elapsed_time(b_) for a, b_ in zip(iso_a, iso_b)) / n_iso
            cands = int(sum(int(c.sum().item()) for c in cnt_t) / n_sets)
            rr_bytes = 4 * args.dim * cands + 8 * cands + 4 * args.dim * nq
            out["rerank"] = {"kernel": "rerank_kernel", "kernel_ms": round(rr_ms, 4), "launches": n_iso,
                             "kernel_ms_inside_the_pipeline": round(rr_pipe_ms, 4),
                             "candidates": cands, "algorithmic_bytes": rr_bytes,
                             "achieved_GBps": round(rr_bytes / (rr_ms * 1e-3) / 1e9, 1),
                             "frac_of_hbm_peak": round(rr_bytes / (rr_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                             "note": "BM25 top-1000 candidates reranked by cosine (dot of unit vectors) "
                                     "with alpha=0.5 blend to top-10; value = whole pipeline"}
            # BASELINE.json labels config 5 "(MFMA path)": the same device candidates through the
            # 2-clause cosine rerank, whose [candidates x clauses] products run on v_mfma_f32_16x16x4_f32
            q2 = torch.from_numpy(np.stack([corpus.unit_vectors(nq, args.dim, seed=12),
                                            corpus.unit_vectors(nq, args.dim, seed=13)], axis=1)).cuda()
            a2 = torch.full((nq, 2), 0.5, dtype=torch.float32, device="cuda")
            n_mf = max(4, min(args.steps, 16))
            me_a = [torch.cuda.Event(enable_timing=True) for _ in range(n_mf)]
            me_b = [torch.cuda.Event(enable_timing=True) for _ in range(n_mf)]
            torch.cuda.synchronize()
            for i in range(-2, n_mf):
                d = d_res[i % n_sets]
                if i >= 0:
                    me_a[i].record()
                index.rerank_multi_batch_device(nq, 2, q2.data_ptr(), a2.data_ptr(), None, d[0], d[1], d[2], d[3],
                                                k, k_out, r_doc.data_ptr(), r_seg.data_ptr(), r_score.data_ptr(),
                                                r_vec.data_ptr(), r_count.data_ptr())
                if i >= 0:
                    me_b[i].record()
            torch.cuda.synchronize()
            mf_ms = sum(a.elapsed_time(b_) for a, b_ in zip(me_a, me_b)) / n_mf
            mf_bytes = 4 * args.dim * cands + 8 * cands + 2 * 4 * args.dim * nq
            out["rerank_mfma"] = {"kernel": "rerank_multi_kernel (2 cosine clauses, v_mfma_f32_16x16x4_f32)",
                                  "kernel_ms": round(mf_ms, 4), "launches": n_mf, "clauses": 2,
                                  "algorithmic_bytes": mf_bytes,
                                  "achieved_GBps": round(mf_bytes / (mf_ms * 1e-3) / 1e9, 1),
                                  "frac_of_hbm_peak": round(mf_bytes / (mf_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                  "flops": 2 * 2 * args.dim * cands,
                                  "note": "HBM row-gather bound (SURVEY 8d): every candidate row is read once for "
                                          "both clauses; the matrix cores do the contraction"}

    # ---- parity spot-check + CPU baseline (rank 0; the baseline at N = 1 only) ----
    if rank == 0:
        from oracle import oracle as O
        offs, terms, w = qs[0]
        terms = terms.reshape(-1)
        TQ = T * MF_FIELDS if plans else T  # scored terms per query
        leaf0 = q_leaves[0] if plans else None
        lk = (lambda n: {"q_leaf": leaf0[:n * TQ]}) if plans else (lambda n: {})
        got = batches[0].fetch() if rerank else None
        if not rerank:
            batches[0].run()
            got = batches[0].fetch()
            hp = first_results.get(0)  # (absent with --kernel-leg-only)  # the host-inclusive leg's result for the same query set
            if hp is not None and not all(np.array_equal(np.asarray(x).view(np.uint32), np.asarray(y).view(np.uint32))
                                          for x, y in zip(got[:4], hp[:4])):
                raise SystemExit("bench.py: host-inclusive and device-resident results disagree")
        nchk = min(args.check, nq)
        if nchk:
            want = O.search_batch([seg], offs[:nchk + 1], terms[:nchk * TQ], w[:nchk * TQ], k,
                                  strategy=O.BM25, n_threads=gen_threads, **lk(nchk))
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
            # rerank spot check against the oracle (tolerance 1e-5 on blended scores): re-run set 0
            batches[0].run()
            batches[0].rerank_device(1, qv.data_ptr(), alpha.data_ptr(), None, k_out, r_doc.data_ptr(), r_seg.data_ptr(),
                                     r_score.data_ptr(), r_vec.data_ptr(), r_count.data_ptr())
            torch.cuda.synchronize()
            got = batches[0].fetch()
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
            co, ct, cw = offs[:ncpu + 1], terms[:ncpu * TQ], w[:ncpu * TQ]
            ostrat = {"bm25": O.BM25, "wand": O.WAND, "bmw": O.BMW}[args.strategy]
            # strict baseline: scorer only, min_doc_len cached (a cache the reference lacks)
            reps_c, t_cpu = 0, 0.0
            while t_cpu < 10.0 and reps_c < 50:
                tc = time.perf_counter()
                O.search_batch([seg], co, ct, cw, k, strategy=ostrat, n_threads=cores, cache_min_len=True, **lk(ncpu))
                t_cpu += time.perf_counter() - tc
                reps_c += 1
            strict = ncpu * reps_c / t_cpu
            # faithful: TermState::new rescans all doc lengths per term per query (wand.rs:111-125)
            # (the whole CPU sample, repeated until >= 2 s have passed: 256 queries on 256 threads once was a
            #  single wave of work whose time was mostly thread start-up)
            nf = min(ncpu, 1024)
            reps_f, t_f = 0, 0.0
            while t_f < 2.0 and reps_f < 20:
                tc = time.perf_counter()
                O.search_batch([seg], offs[:nf + 1], terms[:nf * TQ], w[:nf * TQ], k, strategy=ostrat,
                               n_threads=cores, cache_min_len=False, **lk(nf))
                t_f += time.perf_counter() - tc
                reps_f += 1
            faithful = nf * reps_f / t_f
            # BASELINE.md "Baseline A" (context): + per-query varint decode of every list (twice) and
            # the O(N) doc-length rebuild IndexReader::search does around the scorer
            na = min(ncpu, 4 * cores)
            secs_a = None
            if not plans:  # (the Baseline A restatement takes flat queries)
                _, secs_a = O.search_batch_faithful([seg], offs[:na + 1], terms[:na * T], w[:na * T], k,
                                                    strategy=ostrat, n_threads=cores)
            # Baseline B is `wand` for config 2 and `bmw` (block 128, postings.rs:11) for
            # config 3; the other pruned strategy is timed beside it on the same sample
            other = O.BMW if ostrat != O.BMW else O.WAND
            reps_o, t_o = 0, 0.0
            while t_o < 4.0 and reps_o < 20:
                tc = time.perf_counter()
                O.search_batch([seg], co, ct, cw, k, strategy=other, block_size=128, n_threads=cores,
                               cache_min_len=True, **lk(ncpu))
                t_o += time.perf_counter() - tc
                reps_o += 1
            other_rate = ncpu * reps_o / t_o
            out["cpu_baseline"] = {
                "value": round(strict, 1), "unit": "queries/s", "cores": cores, "kind": "port",
                "visible_cpus": visible_cpus, "cgroup_cpu_quota": cpu_quota,
                ("bmw_block128_value" if other == O.BMW else "wand_value"): round(other_rate, 1),
                "sample": f"{ncpu} queries of query set 0 x {reps_c} reps, oracle "
                          f"{args.strategy} (C restatement of searchlite-core's scorer, "
                          f"pre-decoded postings, cached doc lengths and min_doc_len), "
                          f"{cores} threads, one query per thread",
                "faithful_value": round(faithful, 1),
                "faithful_note": "same, but with the reference's per-term O(N) min_doc_len scan "
                                 f"(wand.rs:111-125) on {nf} queries x {reps_f} reps",
                "baseline_a_value": None if secs_a is None else round(na / secs_a, 1),
                "baseline_a_note": "BASELINE.md Baseline A on the same cores: per query every term's list is "
                                   "varint-decoded twice from its serialized form and the dense doc-length "
                                   f"vector is rebuilt (api/reader.rs:1732-1735, 3604-3621), then wand; {na} queries",
                "gpu_over_cpu": round(value / strict, 1)}

    # ---- request coalescer (N = 1, config 2): single-query callers, one blocking thread each ----
    if rank == 0 and world == 1 and args.config == "c2" and args.coalesce_threads > 0 and not args.kernel_leg_only:
        import ctypes as C
        from searchlite_amd import build as sbuild
        Lh = C.CDLL(sbuild.build_harness())
        Lh.slh_coalesce_bench.restype = C.c_double
        Lh.slh_coalesce_bench.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        offs, terms, w = (np.ascontiguousarray(x) for x in qs[0])
        batches[0].run()
        exp = batches[0].fetch()
        e_doc, e_score, e_cnt = (np.ascontiguousarray(exp[0], np.uint32), np.ascontiguousarray(exp[2], np.float32),
                                 np.ascontiguousarray(exp[3], np.uint32))
        legs = []
        # (threads, requests in flight per thread): blocking callers, then callers that pipeline (submit / wait)
        for thr, depth, wait_us in ((16, 1, 30), (64, 1, 30), (args.coalesce_threads, 1, 30),
                                    (4 * args.coalesce_threads, 1, 30), (16, 64, 30), (16, 256, 30), (8, 256, 30)):
            bad, nb = C.c_int64(0), C.c_uint64(0)
            ph = (C.c_double * 4)()
            total = 64 * nq
            Lh.slh_coalesce_bench(index._h, local_rank, thr, 8 * nq, offs.ctypes.data, terms.ctypes.data, w.ctypes.data,
                                  nq, 1, k, strategy, 1024, wait_us, None, None, None, None, None, None, depth)  # warm-up
            secs = Lh.slh_coalesce_bench(index._h, local_rank, thr, total, offs.ctypes.data, terms.ctypes.data,
                                         w.ctypes.data, nq, 1, k, strategy, 1024, wait_us, e_doc.ctypes.data,
                                         e_score.ctypes.data, e_cnt.ctypes.data, C.addressof(bad), C.addressof(nb), ph,
                                         depth)
            if secs <= 0:
                raise SystemExit("bench.py: the coalescer leg failed")
            legs.append({"caller_threads": thr, "in_flight_per_thread": depth, "max_wait_us": wait_us, "queries": total,
                         "queries_per_s": round(total / secs, 1), "batches": int(nb.value),
                         "mean_batch": round(total / max(1, nb.value), 1),
                         "leader_ms_per_batch": {"collect": round(ph[0], 3), "prepare": round(ph[1], 3),
                                                 "run": round(ph[2], 3), "fetch_destroy": round(ph[3], 3)},
                         "rows_differing_from_the_batch_api": int(bad.value)})
            if bad.value:
                print(json.dumps(legs))
                raise SystemExit("bench.py: coalescer results differ from the batch API")
        out["config"]["coalescer"] = {
            "is": "in_flight_per_thread 1: slg_coalescer_search, every caller thread blocks with ONE query (the reference "
                  "serves a request per blocking thread, searchlite-http/src/lib.rs:628-652); > 1: slg_coalescer_submit / "
                  "_wait, a thread keeps that many requests in flight; concurrent requests are collected into batches "
                  "behind the C ABI; every row compared bit for bit with the batch API's",
            "visible_cpus": visible_cpus, "cgroup_cpu_quota": cpu_quota,
            "bound": "T blocking callers = T queries outstanding: rate <= T / (latency of a batch from its first row "
                     "to its callers' wake-up) (Little's law)",
            "legs": legs}

    for b in batches:
        b.close()
    index.close()
    del seg

    # ---- N > 1: the BASELINE configs[3] shape, index-sharded (strong scaling) ----
    if world > 1 and not args.no_c4 and not rerank:
        c4, _, _, _ = run_sharded("c4", max(4, min(args.steps, 16)), max(1, min(args.warmup, 3)))
        if rank == 0:
            out["c4"] = c4
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
