"""The request coalescer (slg_coalescer_*): concurrent single-query callers — the reference serves
every request on its own blocking thread (searchlite-http/src/lib.rs:628-652) and has no batch API
(api/reader.rs:2539) — are collected into batches behind the C ABI; every caller gets the row the
batch API returns for the same query, bit for bit."""
import ctypes as C
import threading

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


def harness():
    from searchlite_amd import build as sbuild
    L = C.CDLL(sbuild.build_harness())
    L.slh_coalesce_bench.restype = C.c_double
    L.slh_coalesce_bench.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    return L


@pytest.mark.parametrize("k,threads,max_batch,depth", [(11, 64, 1024, 1), (11, 7, 4, 1), (300, 32, 16, 1),
                                                        (11, 4, 1024, 32), (11, 3, 8, 5)])
def test_concurrent_callers_get_the_batch_api_rows(gpu, oracle, k, threads, max_batch, depth):
    """depth 1: blocking callers (slg_coalescer_search); depth > 1: every thread keeps that many tickets in
    flight (slg_coalescer_submit / slg_coalescer_wait)."""
    rng = np.random.default_rng(40 + k + threads)
    segs = [random_segment(rng, 4000, 80, 25, k1=0.9, b=0.4), random_segment(rng, 2500, 80, 25, k1=0.9, b=0.4)]
    nq = 96
    offs, terms, w = random_queries(rng, nq, 3, 80, n_segs=2, weights=True)
    want = oracle.search_batch(segs, offs, terms, w, k, strategy=oracle.BM25)
    L = harness()
    with gpu.GpuIndex(segs) as ix:
        exp = ix.search_batch(offs, terms, w, k, gpu.Wand)
        assert_same_hits(exp, want, 0.0, "batch API")
        e_doc, e_score, e_cnt = (np.ascontiguousarray(exp[0], np.uint32), np.ascontiguousarray(exp[2], np.float32),
                                 np.ascontiguousarray(exp[3], np.uint32))
        offs_c, terms_c, w_c = (np.ascontiguousarray(offs, np.uint32), np.ascontiguousarray(terms, np.uint32),
                                np.ascontiguousarray(w, np.float32))
        bad, nb = C.c_int64(-1), C.c_uint64(0)
        secs = L.slh_coalesce_bench(ix._h, 0, threads, 20 * nq, offs_c.ctypes.data, terms_c.ctypes.data,
                                    w_c.ctypes.data, nq, 2, k, gpu.Wand, max_batch, 50, e_doc.ctypes.data,
                                    e_score.ctypes.data, e_cnt.ctypes.data, C.addressof(bad), C.addressof(nb), None, depth)
        assert secs > 0 and bad.value == 0
        assert 1 <= nb.value <= 20 * nq
        if threads >= 32 and max_batch >= threads:
            assert nb.value < 20 * nq  # callers really shared batches


def test_lone_caller_and_mixed_k(gpu, oracle):
    """A lone request does not wait for company; callers with different k never share a batch."""
    from searchlite_amd import _native as N
    rng = np.random.default_rng(9)
    seg = random_segment(rng, 3000, 50, 20, k1=0.9, b=0.4)
    offs, terms, w = random_queries(rng, 8, 3, 50, weights=True)
    lib = N.load()
    with gpu.GpuIndex([seg]) as ix:
        co = lib.slg_coalescer_create(ix._h, 64, 20000)  # a 20 ms wait would show
        assert co
        results = {}

        def one(q, k):
            tid = np.ascontiguousarray(terms[offs[q]:offs[q + 1]].reshape(-1), np.uint32)
            ww = np.ascontiguousarray(w[offs[q]:offs[q + 1]], np.float32)
            qq = N.Query(len(ww), tid.ctypes.data, ww.ctypes.data)
            d, s, sc, c = np.zeros(k, np.uint32), np.zeros(k, np.uint32), np.zeros(k, np.float32), C.c_uint32(0)
            rc = lib.slg_coalescer_search(co, C.addressof(qq), k, gpu.Wand, d.ctypes.data, s.ctypes.data,
                                          sc.ctypes.data, C.addressof(c), None)
            results[(q, k)] = (rc, d, sc, c.value)

        import time
        t0 = time.perf_counter()
        one(0, 11)
        assert time.perf_counter() - t0 < 0.015  # idle coalescer: no collection wait
        th = [threading.Thread(target=one, args=(q, 11 if q % 2 else 30)) for q in range(8)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        lib.slg_coalescer_destroy(co)
        for k in (11, 30):
            want = oracle.search_batch([seg], offs, terms, w, k, strategy=oracle.BM25)
            for q in range(8):
                if (q, k) not in results:
                    continue
                rc, d, sc, c = results[(q, k)]
                assert rc == 0 and c == int(want[3][q])
                assert np.array_equal(d[:c], want[0][q, :c])
                assert np.array_equal(sc[:c].view(np.uint32), want[2][q, :c].view(np.uint32))


def test_tickets_one_thread_many_requests(gpu, oracle):
    """slg_coalescer_submit / _poll / _wait: ONE thread keeps 40 requests in flight, collects them in
    another order than it submitted them; a ticket is good for one wait; a wait without output
    arrays fails but gives its row back (the batch object is recycled: a second round works)."""
    from searchlite_amd import _native as N
    rng = np.random.default_rng(19)
    seg = random_segment(rng, 5000, 60, 22, k1=0.9, b=0.4)
    nq, k = 40, 11
    offs, terms, w = random_queries(rng, nq, 3, 60, weights=True)
    want = oracle.search_batch([seg], offs, terms, w, k, strategy=oracle.BM25)
    lib = N.load()
    with gpu.GpuIndex([seg]) as ix:
        co = lib.slg_coalescer_create(ix._h, 16, 50)  # (batches of at most 16: the 40 requests span several)
        assert co
        for round_ in range(2):
            keep, tickets = [], []
            for q in range(nq):
                tid = np.ascontiguousarray(terms[offs[q]:offs[q + 1]].reshape(-1), np.uint32)
                ww = np.ascontiguousarray(w[offs[q]:offs[q + 1]], np.float32)
                qq = N.Query(len(ww), tid.ctypes.data, ww.ctypes.data)
                t = N.Ticket()
                assert lib.slg_coalescer_submit(co, C.addressof(qq), None, 0, 0.0, 0, -1, k, gpu.Wand, 0,
                                                C.addressof(t)) == 0
                tickets.append(t)
                del tid, ww  # (the query's arrays may be reused as soon as submit returns)
            for q in reversed(range(nq)):
                t = tickets[q]
                d, s, sc, c = np.zeros(k, np.uint32), np.zeros(k, np.uint32), np.zeros(k, np.float32), C.c_uint32(0)
                if q == 7 and round_ == 0:  # no output arrays: an error, and the row is given back
                    assert lib.slg_coalescer_wait(co, C.addressof(t), None, None, None, None, None) == N.ERR_INVALID
                    assert not t.batch
                    continue
                rc = lib.slg_coalescer_wait(co, C.addressof(t), d.ctypes.data, s.ctypes.data, sc.ctypes.data,
                                            C.addressof(c), None)
                assert rc == 0 and not t.batch
                assert lib.slg_coalescer_poll(co, C.addressof(t)) == N.ERR_INVALID  # a ticket is good for one wait
                assert lib.slg_coalescer_wait(co, C.addressof(t), d.ctypes.data, s.ctypes.data, sc.ctypes.data,
                                              C.addressof(c), None) == N.ERR_INVALID
                n = int(want[3][q])
                assert c.value == n and np.array_equal(d[:n], want[0][q, :n])
                assert np.array_equal(sc[:n].view(np.uint32), want[2][q, :n].view(np.uint32))
        nb, nqs = C.c_uint64(0), C.c_uint64(0)
        assert lib.slg_coalescer_stats(co, C.addressof(nb), C.addressof(nqs)) == 0
        assert nqs.value == 2 * nq and nb.value >= 2 * 3  # 40 requests in batches of at most 16
        lib.slg_coalescer_destroy(co)
