"""Index-sharded runs with several caller threads per rank (world 1 on one GPU; RCCL behind the C ABI).

Collectives on one communicator must be issued in the same order on every rank: the shard group issues
them on a stream of its own in the order of the runs' sequence numbers (slg_batch_run_sharded_seq), so
caller threads may prepare and launch their batches concurrently — which is how bench.py's config-4
leg keeps host planning off the critical path (searchlite_amd/csrc/tools/host_harness.cpp)."""
import ctypes as C
import threading
import time

import numpy as np
import pytest

from tests.util import assert_same_hits, random_queries, random_segment

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import searchlite_amd as sa
    from searchlite_amd import searcher
    assert searcher.device_count() >= 1
    return sa


def test_harness_keeps_sharded_batches_in_flight(gpu, oracle):
    """Three native caller threads, each with two sharded batches in flight on streams of its own,
    twelve steps over two query sets: every result equals the oracle's merged top-k."""
    from searchlite_amd import build as sbuild, searcher
    rng = np.random.default_rng(61)
    vocab, k, nq = 70, 11, 64
    segs = [random_segment(rng, 3000 + 500 * i, vocab, 25, k1=0.9, b=0.4) for i in range(2)]
    sets = [random_queries(rng, nq, 3, vocab, n_segs=2, weights=True) for _ in range(2)]
    want = [oracle.search_batch(segs, o, t, w, k, strategy=oracle.BM25) for o, t, w in sets]
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
    keep = [(np.ascontiguousarray(o, np.uint32), np.ascontiguousarray(t, np.uint32), np.ascontiguousarray(w, np.float32))
            for o, t, w in sets]
    ptrs = lambda j: (C.c_void_p * len(keep))(*[x[j].ctypes.data for x in keep])
    with gpu.GpuIndex(segs) as ix:
        ix.profile(True)
        group = searcher.ShardGroup(ix, 0, 1, searcher.shard_unique_id(), 2)
        h = L.slh_create(ix._h, 0, 3, len(keep), ptrs(0), ptrs(1), ptrs(2), nq, k, gpu.Wand)
        L.slh_set_group(h, group._h)
        assert L.slh_run(h, 0, 5) == 0, L.slh_error(h)
        assert L.slh_run(h, 5, 7) == 0, L.slh_error(h)  # (step numbers go on: the collective order is 0, 1, 2, ...)
        for j in range(2):
            d = np.zeros((nq, k), np.uint32)
            s = np.zeros((nq, k), np.uint32)
            sc = np.zeros((nq, k), np.float32)
            c = np.zeros(nq, np.uint32)
            assert L.slh_first_result(h, j, d.ctypes.data, s.ctypes.data, sc.ctypes.data, c.ctypes.data) == 1
            assert_same_hits((d, s, sc, c), want[j], 0.0, f"sharded harness, query set {j}")
        L.slh_destroy(h)
        st = group.stats()
        assert st["runs"] == 12 and st["kernel_ms"] > 0 and st["gather_ms"] >= 0 and st["merge_ms"] > 0
        group.close()


def test_collectives_follow_the_sequence_numbers(gpu, oracle):
    """Run 1 is called before run 0: it waits for its turn, run 0 goes first, both return the merged
    result of their own batch."""
    from searchlite_amd import searcher
    from searchlite_amd.searcher import PreparedBatch
    rng = np.random.default_rng(62)
    vocab, k = 50, 11
    seg = random_segment(rng, 2500, vocab, 20, k1=0.9, b=0.4)
    qa = random_queries(rng, 32, 3, vocab, weights=True)
    qb = random_queries(rng, 32, 2, vocab, weights=True)
    want_a = oracle.search_batch([seg], *qa, k, strategy=oracle.BM25)
    want_b = oracle.search_batch([seg], *qb, k, strategy=oracle.BM25)
    with gpu.GpuIndex([seg]) as ix:
        group = searcher.ShardGroup(ix, 0, 1, searcher.shard_unique_id(), 1)
        ba = PreparedBatch(ix, *qa, k, gpu.Wand)
        bb = PreparedBatch(ix, *qb, k, gpu.Wand)
        got = {}
        order = []

        def late():  # sequence number 1, called first
            bb.run_sharded(group, fetch=False, seq=1)
            order.append(1)
            got["b"] = bb.fetch_sharded()

        t = threading.Thread(target=late)
        t.start()
        time.sleep(0.2)
        assert order == []          # still waiting for run 0's collective
        ba.run_sharded(group, fetch=False, seq=0)
        order.append(0)
        got["a"] = ba.fetch_sharded()
        t.join(timeout=30)
        assert not t.is_alive()
        assert_same_hits(got["a"], want_a, 0.0, "run 0")
        assert_same_hits(got["b"], want_b, 0.0, "run 1")
        ba.close()
        bb.close()
        group.close()
