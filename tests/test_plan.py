"""CPU unit tests of the host planner (searchlite_amd/csrc/slg_plan.cpp), through the test C ABI of
lib/libslg_plan.so: the planner is a pure host function (segments' host mirrors + query arrays ->
descriptor image), so its invariants are checked without a GPU: every round of every sub-query is
owned by exactly one slice, launch order is a permutation (most rounds first), cut-point rows fit
the few-term kernel's 64-word row, the threshold seed never exceeds the true k-th score (oracle),
MaxScore's non-essential lists really sum below the seed, malformed input is rejected."""
import ctypes as C
import os

import numpy as np
import pytest

from tests.util import random_queries, random_segment

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RQ = np.dtype([(n, "<u4") for n in ("q", "seg", "term_begin", "n_terms", "slice_begin", "n_slices", "n_rounds",
                                     "rounds_per_slice", "bounds_begin", "rdoc_begin", "bnd_begin", "longest",
                                     "ess_mask", "skip_mask", "filter", "cand_lo", "cand_hi", "plan")]
              + [("tie", "<f4"), ("max_init", "<f4"), ("n_leaves", "<u4"), ("n_groups", "<u4"), ("theta0", "<f4"),
                 ("depth", "<u4"), ("node_begin", "<u4")])
PN = np.dtype([("parent", "<u4"), ("n_children", "<u4"), ("kind", "<u4"), ("tie", "<f4")])
TR = np.dtype([("off", "<u8"), ("df", "<u4"), ("weight", "<f4"), ("term", "<u4"), ("leaf", "<u4"),
               ("gmeta", "<u4"), ("gtie", "<f4")])
KCHAMP = 68


class Seg(C.Structure):
    _fields_ = [("n_docs", C.c_uint32), ("n_terms", C.c_uint32), ("term_offsets", C.c_void_p), ("champ", C.c_void_p)]


class Facts(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_postings", "n_postings_essential", "n_postings_nonessential",
                                          "n_rounds", "n_bounds", "n_bnd", "cand_total", "image_bytes")] + \
               [(n, C.c_uint32) for n in ("n_sq", "n_terms", "n_slices", "max_terms", "uniform", "multi",
                                          "plan_batch", "nested", "pruned", "cand_mode", "sizeof_round_query",
                                          "sizeof_term_ref", "deep")]


@pytest.fixture(scope="module")
def lib():
    from searchlite_amd import build
    L = C.CDLL(build.build_plan_lib())
    L.slgp_plan.restype = C.c_void_p
    L.slgp_plan.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                            C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_uint32, C.c_char_p,
                            C.c_uint32, C.c_void_p]
    L.slgp_facts_of.argtypes = [C.c_void_p, C.c_void_p]
    L.slgp_bytes.restype = C.c_uint64
    L.slgp_bytes.argtypes = [C.c_void_p, C.c_int]
    L.slgp_copy.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.slgp_free.argtypes = [C.c_void_p]
    return L


def default_tuning(**over):
    from searchlite_amd import _native as N
    t = N.Tuning()
    t.struct_size = C.sizeof(N.Tuning)
    t.validate, t.champions, t.pruning = 1, 1, -1
    t.uniform_max_terms, t.multi_round_target, t.probe_target = 8, 448, 2048
    t.slices_per_subquery, t.cand_mode, t.slice_order, t.block_max = 16, 1, 1, 1
    t.uniform_kernel = 4
    t.updatable, t.uniform_plans = 1, 1
    for k, v in over.items():
        setattr(t, k, v)
    return t


def impacts_of(seg):
    """numpy f32 restatement of bm25 (query/bm25.rs:1-6 via score_tf wand.rs:279-285), per posting."""
    f32 = np.float32
    offs = np.asarray(seg.term_offsets, dtype=np.int64)
    df = np.diff(offs).astype(np.float32)
    idf = np.maximum(np.log((f32(seg.docs) - df + f32(0.5)) / (df + f32(0.5))).astype(np.float32), f32(0)) + f32(1)
    term_of = np.repeat(np.arange(len(df)), np.diff(offs))
    tf = seg.tfs.astype(np.float32)
    avgdl = f32(seg.field_avgdl[0])
    dl = seg.field_doc_len[0][seg.doc_ids]
    dl = np.where(dl > 0, dl, np.maximum(avgdl, f32(1)))
    norm = (dl / avgdl).astype(np.float32)
    denom = tf + f32(seg.k1) * (f32(1) - f32(seg.b) + f32(seg.b) * norm)
    return (idf[term_of] * (tf * (f32(seg.k1) + f32(1))) / np.maximum(denom, f32(1e-6))).astype(np.float32)


def champions_of(seg):
    """Exact order statistics as the champion table: champ[t][r] = (r+1)-th largest impact (0 past the
    list), entries 64..67 = the 128 / 256 / 512 / 1024-th largest — valid lower bounds, which is all
    the planner needs (the device computes looser ones)."""
    imp = impacts_of(seg)
    offs = np.asarray(seg.term_offsets, dtype=np.int64)
    V = len(offs) - 1
    ch = np.zeros((V, KCHAMP), dtype=np.float32)
    for t in range(V):
        x = np.sort(imp[offs[t]:offs[t + 1]])[::-1]
        ch[t, :min(64, len(x))] = x[:64]
        for j, r in enumerate((128, 256, 512, 1024)):
            if len(x) >= r:
                ch[t, 64 + j] = x[r - 1]
    return ch


class Planned:
    def __init__(self, lib, segs, offs, terms, w, k, strategy=1, tuning=None, plans=None, q_filter=None,
                 filter_live=b"", champs=None):
        self.lib = lib
        n_segs = len(segs)
        self.keep = []
        arr = (Seg * n_segs)()
        for i, s in enumerate(segs):
            to = np.ascontiguousarray(s.term_offsets, dtype=np.uint64)
            ch = None if champs is None else np.ascontiguousarray(champs[i], dtype=np.float32)
            self.keep += [to, ch]
            arr[i] = Seg(s.n_docs, s.n_terms, to.ctypes.data, None if ch is None else ch.ctypes.data)
        offs = np.ascontiguousarray(offs, dtype=np.uint32)
        terms = np.ascontiguousarray(terms, dtype=np.uint32)
        w = np.ascontiguousarray(w, dtype=np.float32)
        tune = tuning or default_tuning()
        err = C.create_string_buffer(512)
        code = C.c_int(0)
        qf = None if q_filter is None else np.ascontiguousarray(q_filter, dtype=np.int32)
        self.h = lib.slgp_plan(C.addressof(arr), n_segs, C.addressof(tune), len(offs) - 1, offs.ctypes.data,
                               terms.ctypes.data, w.ctypes.data, None if plans is None else C.addressof(plans),
                               None if qf is None else qf.ctypes.data, k, strategy, filter_live, len(filter_live),
                               err, 512, C.addressof(code))
        self.code, self.err = code.value, err.value.decode()
        if self.h:
            self.facts = Facts()
            lib.slgp_facts_of(self.h, C.addressof(self.facts))

    def array(self, what, dtype):
        n = self.lib.slgp_bytes(self.h, what)
        buf = np.zeros(n, dtype=np.uint8)
        if n:
            self.lib.slgp_copy(self.h, what, buf.ctypes.data)
        return buf.view(dtype)

    def close(self):
        if self.h:
            self.lib.slgp_free(self.h)
            self.h = None


def check_structure(p, k):
    f = p.facts
    assert f.sizeof_round_query == RQ.itemsize and f.sizeof_term_ref == TR.itemsize
    sqs, terms = p.array(0, RQ), p.array(1, TR)
    slice_sq, slice_seg, order = p.array(2, "<u4"), p.array(3, "<u4"), p.array(4, "<u4")
    assert len(sqs) == f.n_sq and len(terms) == f.n_terms and len(slice_sq) == f.n_slices
    # launch order: a permutation of the slices, most rounds first
    assert sorted(order.tolist()) == list(range(f.n_slices))
    rounds_of = np.zeros(f.n_slices, dtype=np.int64)
    next_slice, next_bounds, next_bnd, next_term = 0, 0, 0, 0
    for i, sq in enumerate(sqs):
        T, nr, rps, S = int(sq["n_terms"]), int(sq["n_rounds"]), int(sq["rounds_per_slice"]), int(sq["n_slices"])
        assert T >= 1 and nr >= 1 and rps >= 1 and S == (nr + rps - 1) // rps
        # sub-queries own consecutive, non-overlapping ranges of slices, terms, cut points, boundaries
        assert sq["slice_begin"] == next_slice and sq["term_begin"] == next_term
        assert sq["bounds_begin"] == next_bounds and sq["bnd_begin"] == next_bnd == sq["rdoc_begin"]
        next_slice += S
        next_term += T
        next_bounds += (nr + 1) * T
        next_bnd += nr + 1
        # every round in exactly one slice
        covered = 0
        for j in range(S):
            s = int(sq["slice_begin"]) + j
            assert slice_sq[s] == i and slice_seg[s] == sq["seg"]
            rounds_of[s] = min(rps, nr - j * rps)
            assert rounds_of[s] >= 1
            covered += rounds_of[s]
        assert covered == nr
        if f.uniform:
            assert (rps + 1) * T <= (128 if T > 4 else 64) and T <= 8     # the slice's cut points fit one row
            assert rps <= (16 if f.max_terms > 4 else 8)
        tt = terms[int(sq["term_begin"]):int(sq["term_begin"]) + T]
        assert (np.diff(tt["leaf"].astype(np.int64)) >= 0).all()          # lists sorted by leaf
        ess = int(sq["ess_mask"])
        assert ess & ((1 << T) - 1) and (ess >> int(sq["longest"])) & 1   # >= 1 essential list; the splitter is one
        assert nr <= int(tt["df"][int(sq["longest"])])                    # a round holds >= 1 posting of the splitter
        assert (int(sq["skip_mask"]) & ess) == 0
    assert next_slice == f.n_slices and next_bounds == f.n_bounds and next_bnd == f.n_bnd
    r_in_order = rounds_of[order]
    assert (np.diff(r_in_order) <= 0).all(), "slices must launch in non-increasing order of rounds"
    assert int(rounds_of.sum()) == f.n_rounds
    assert int(p.array(7, "<u8").sum()) == f.n_postings
    if not f.cand_mode:
        assert f.n_slices * max(k, 1) < 2 ** 32
    # the packed image holds the arrays where the layout says
    img = p.array(8, np.uint8)
    assert len(img) == f.image_bytes and f.image_bytes % 16 == 0
    assert img[:len(sqs) * RQ.itemsize].tobytes() == sqs.tobytes()
    return sqs, terms


@pytest.mark.parametrize("T,k,n_segs", [(3, 11, 1), (2, 11, 2), (5, 101, 1), (4, 1001, 1), (13, 11, 2), (32, 5, 1)])
def test_round_and_slice_invariants(lib, T, k, n_segs):
    rng = np.random.default_rng(100 + T)
    segs = [random_segment(rng, 4000 + 500 * s, 60, 25) for s in range(n_segs)]
    offs, terms, w = random_queries(rng, 40, T, 60, n_segs=n_segs, weights=True)
    champs = [champions_of(s) for s in segs]
    for strat in (0, 1):
        p = Planned(lib, segs, offs, terms, w, k, strategy=strat, champs=champs)
        assert p.h, p.err
        sqs, tr = check_structure(p, k)
        assert p.facts.uniform == (T <= 8)
        assert p.facts.cand_mode == (k > 256)
        want_postings = sum(int(s.df(int(terms[i, j]))) for i in range(terms.shape[0]) for j, s in enumerate(segs))
        assert p.facts.n_postings == want_postings
        # padded layout: the device offset of term t is term_offsets[t] + 64 * t
        for sq in sqs[:10]:
            for t in tr[int(sq["term_begin"]):int(sq["term_begin"]) + int(sq["n_terms"])]:
                s = segs[int(sq["seg"])]
                assert int(t["off"]) == int(s.term_offsets[int(t["term"])]) + 64 * int(t["term"])
                assert int(t["df"]) == s.df(int(t["term"]))
        p.close()


def test_kernel_choice_follows_what_block_skipping_can_save(lib):
    """5-term Wand batches are MaxScore-classified, and keep the classification (many-term kernel,
    block skipping) only where skipping is expected to leave >= 15 % of the postings unread: lists of
    similar density (config 3's shape) go to the few-term kernel unclassified; a stop word next to
    rare terms stays classified.  pruning = 1 / 0 force either way; the slot form of the few-term
    kernel plans smaller rounds than the blocked form."""
    from searchlite_amd import corpus
    seg = corpus.zipf_segment(200_000, 1 << 14, seed=5)
    ch = champions_of(seg)
    offs, terms, w = corpus.zipf_queries(64, 5, rank_lo=8, rank_hi=2048, seed=3, vocab=1 << 14)
    p = Planned(lib, [seg], offs, terms, w, 101, strategy=1, champs=[ch])
    assert p.h, p.err
    sqs, _ = check_structure(p, 101)
    assert p.facts.uniform and not p.facts.pruned and p.facts.n_postings_nonessential == 0
    assert all(int(sq["ess_mask"]) == 31 and int(sq["skip_mask"]) == 0 for sq in sqs)
    rounds_blocked = p.facts.n_rounds
    p.close()
    forced = Planned(lib, [seg], offs, terms, w, 101, strategy=1, champs=[ch], tuning=default_tuning(pruning=1))
    check_structure(forced, 101)
    assert forced.facts.pruned and forced.facts.multi and forced.facts.n_postings_nonessential > 0
    forced.close()
    slots = Planned(lib, [seg], offs, terms, w, 101, strategy=1, champs=[ch], tuning=default_tuning(uniform_kernel=3))
    check_structure(slots, 101)
    assert slots.facts.uniform and slots.facts.n_rounds > 1.2 * rounds_blocked
    slots.close()
    # one stop-word-like term (ranks 1..4) among four rare ones
    rng = np.random.default_rng(4)
    t2 = np.stack([np.concatenate([rng.integers(0, 4, 1), rng.choice(np.arange(3000, 12000), 4, replace=False)])
                   for _ in range(64)]).astype(np.uint32)
    for row in t2:
        rng.shuffle(row)
    skew = Planned(lib, [seg], offs, t2.reshape(-1), w, 11, strategy=1, champs=[ch])
    assert skew.h, skew.err
    check_structure(skew, 11)
    assert skew.facts.pruned and skew.facts.multi and any(int(sq["skip_mask"]) for sq in skew.array(0, RQ))
    skew.close()
    off = Planned(lib, [seg], offs, t2.reshape(-1), w, 11, strategy=1, champs=[ch], tuning=default_tuning(pruning=0))
    assert off.facts.uniform and not off.facts.pruned
    off.close()


def test_blocked_round_target_is_the_largest_that_fits(lib):
    """The few-term planner's round target (closed form + correction, slg_plan.cpp) is the largest R in
    steps of 8 whose expected lanes + 3.2 sigma stay under 64.3 — checked against a plain scan over
    all candidates, through the rounds the planner reports (rounds = ceil(postings / R))."""
    import math
    rng = np.random.default_rng(21)
    seg = random_segment(rng, 60000, 50, 40, zipf=True)
    offs, terms, w = random_queries(rng, 200, 5, 50)
    for T in (2, 3, 5, 8):
        o = (np.arange(201) * T).astype(np.uint32)
        t = np.stack([rng.choice(50, size=T, replace=False) for _ in range(200)]).astype(np.uint32).reshape(-1)
        ww = np.ones(200 * T, np.float32)
        p = Planned(lib, [seg], o, t, ww, 11, strategy=0)
        assert p.h, p.err
        sqs, tr = check_structure(p, 11)
        assert p.facts.uniform
        for sq in sqs:
            tt = tr[int(sq["term_begin"]):int(sq["term_begin"]) + int(sq["n_terms"])]
            dfs = [int(x) for x in tt["df"]]
            P, n, lg = sum(dfs), len(dfs), int(sq["longest"])
            def lanes(R):
                mu = var = 0.0
                for j, df in enumerate(dfs):
                    c = R * df / P
                    if j == lg:
                        mu += math.ceil(c / 8.0)
                    else:
                        mu += c / 8.0 + 7.0 / 16.0
                        var += c / 64.0 + 1.0 / 12.0
                return mu + 3.2 * math.sqrt(var)
            best = 512 if n == 1 else max([R for R in range(64, 513, 8) if lanes(R) <= 64.3] or [64])
            best = max(48, min(best, 512))
            want = max(1, min(-(-P // best), dfs[lg]))
            assert int(sq["n_rounds"]) == want, (dfs, int(sq["n_rounds"]), want, best)
        p.close()


def test_threshold_seed_never_exceeds_the_true_kth_score(lib, oracle):
    """theta0 = max_t w_t * champ[t][rank(k)] must be a lower bound of the k-th best score of the
    sub-query (RoundQuery::theta0): checked against the exhaustive oracle; and MaxScore's
    non-essential lists must sum (by their maxima) below it."""
    rng = np.random.default_rng(7)
    seg = random_segment(rng, 6000, 80, 30)
    ch = champions_of(seg)
    for T, k in ((3, 11), (5, 31), (2, 101), (8, 64)):
        offs, terms, w = random_queries(rng, 48, T, 80, weights=True)
        p = Planned(lib, [seg], offs, terms, w, k, strategy=1, champs=[ch], tuning=default_tuning(pruning=1))
        assert p.h, p.err
        sqs, tr = check_structure(p, k)
        want = oracle.search_batch([seg], offs, terms, w, k, strategy=oracle.BM25)
        seeded = 0
        for sq in sqs:
            q = int(sq["q"])
            th = float(sq["theta0"])
            if th > 0:
                seeded += 1
                assert int(want[3][q]) == k, "a seed promises k docs at or above it"
                assert th <= float(want[2][q, k - 1]) * (1 + 1e-6)
            tt = tr[int(sq["term_begin"]):int(sq["term_begin"]) + int(sq["n_terms"])]
            ess = int(sq["ess_mask"])
            noness = [i for i in range(len(tt)) if not (ess >> i) & 1]
            if noness:
                ub = sum(float(tt["weight"][i]) * float(ch[int(tt["term"][i]), 0]) for i in noness)
                assert th > 0 and ub < th
        assert seeded > 0
        # Bm25 (exhaustive strategy) never classifies
        p0 = Planned(lib, [seg], offs, terms, w, k, strategy=0, champs=[ch], tuning=default_tuning(pruning=1))
        assert not p0.facts.pruned
        p0.close()
        p.close()


def test_negative_weights_and_filters_get_no_seed(lib):
    rng = np.random.default_rng(8)
    seg = random_segment(rng, 3000, 40, 20)
    ch = champions_of(seg)
    offs, terms, w = random_queries(rng, 8, 3, 40)
    w[1] = -0.5
    qf = np.array([-1, -1, 0, 0, -1, -1, -1, -1], dtype=np.int32)
    p = Planned(lib, [seg], offs, terms, w, 11, champs=[ch], q_filter=qf, filter_live=b"\x01")
    assert p.h, p.err
    sqs = p.array(0, RQ)
    assert float(sqs[0]["theta0"]) == 0.0                       # negative weight in query 0
    assert float(sqs[2]["theta0"]) == 0.0 and sqs[2]["filter"] == 1   # filtered
    assert float(sqs[4]["theta0"]) > 0.0
    p.close()
    bad = Planned(lib, [seg], offs, terms, w, 11, champs=[ch], q_filter=qf, filter_live=b"\x00")
    assert not bad.h and bad.code == -1 and "filter" in bad.err


def test_malformed_input_is_rejected(lib):
    rng = np.random.default_rng(9)
    seg = random_segment(rng, 500, 40, 10)
    offs, terms, w = random_queries(rng, 4, 3, 40)
    bad_offs = offs.copy()
    bad_offs[2] = 1                                             # not monotone
    assert Planned(lib, [seg], bad_offs, terms, w, 11).code == -1
    far = offs.copy()
    far[1] = 10 ** 6                                            # beyond q_offsets[nq]: must not be followed
    assert Planned(lib, [seg], far, terms, w, 11).code == -1
    o33 = np.array([0, 33], dtype=np.uint32)
    assert Planned(lib, [seg], o33, np.arange(33, dtype=np.uint32), np.ones(33, np.float32), 5).code == -4
    assert Planned(lib, [seg], offs, terms, w, 20002).code == -4     # k > SLG_MAX_K
    t2 = terms.copy()
    t2[0] = 4000
    assert Planned(lib, [seg], offs, t2, w, 11).code == -1           # term id out of range
    w2 = w.copy()
    w2[3] = np.inf
    assert Planned(lib, [seg], offs, terms, w2, 11).code == -1
    assert Planned(lib, [seg], offs, terms, w, 11, strategy=7).code == -1
    ok = Planned(lib, [seg], offs, terms, w, 0)                      # k == 0: no work, no sub-queries
    assert ok.h and ok.facts.n_sq == 0
    ok.close()


def _plans(**kw):
    from searchlite_amd import _native as N
    pl = N.ScorePlans()
    keep = []
    for name, (arr, dt) in kw.items():
        a = np.ascontiguousarray(arr, dtype=dt)
        keep.append(a)
        setattr(pl, name, a.ctypes.data)
    pl._keep = keep
    return pl


def test_two_level_plans_are_validated_and_classified(lib):
    rng = np.random.default_rng(10)
    seg = random_segment(rng, 2000, 40, 15)
    nq, T = 2, 6
    offs, terms, w = random_queries(rng, nq, T, 40)
    leaf = np.tile(np.array([0, 0, 1, 2, 2, 3], dtype=np.uint32), nq)           # 4 leaves
    base = dict(q_leaf=(leaf, "<u4"), q_plan=([1, 0], "<i4"), q_tie=([0.3, 0.0], "<f4"), q_nleaves=([4, 4], "<u4"),
                q_leaf_offsets=([0, 4, 8], "<u4"), q_group_offsets=([0, 2, 4], "<u4"))
    pl = _plans(**base, leaf_group=([0, 0, 1, 1, 0, 1, 1, 1], "<u4"), group_plan=([0, 1, 1, 0], "<i4"),
                group_tie=([0.0, 0.5, 1.0, 0.0], "<f4"))
    p = Planned(lib, [seg], offs, terms, w, 11, plans=pl)
    assert p.h, p.err
    sqs, tr = check_structure(p, 11)
    assert p.facts.nested and p.facts.plan_batch and not p.facts.uniform and not p.facts.pruned
    assert [int(x) for x in sqs["n_groups"]] == [2, 2]
    t0 = tr[:int(sqs[0]["n_terms"])]
    assert [int(g) & 0xFF for g in t0["gmeta"]] == [0, 0, 0, 1, 1, 1]          # group of every list's leaf
    assert [(int(g) >> 8) & 0xFF for g in t0["gmeta"]] == [2, 2, 2, 2, 2, 2]     # leaves the plan gives the group
    assert [(int(g) >> 16) & 1 for g in t0["gmeta"]] == [0, 0, 0, 1, 1, 1]      # group 1 of query 0 is a DisMax
    assert float(t0["gtie"][3]) == 0.5
    p.close()
    # every leaf its own Sum group = the flat plan: not "nested"
    flat = _plans(**base, leaf_group=([0, 1, 2, 3, 0, 1, 2, 3], "<u4"), group_plan=([0] * 8, "<i4"),
                  group_tie=([0.0] * 8, "<f4"))
    flat.q_group_offsets = np.array([0, 4, 8], dtype=np.uint32).ctypes.data
    pf = Planned(lib, [seg], offs, terms, w, 11, plans=flat)
    assert pf.h and not pf.facts.nested and pf.facts.plan_batch   # (two terms share leaf 0: still a plan)
    pf.close()
    # a gap in the groups, a decreasing group, a tie outside [0, 1]
    for lg, gp, gt in (([0, 0, 2, 2, 0, 1, 1, 1], [0, 1, 1, 0], [0, .5, 1, 0]),
                       ([0, 1, 0, 1, 0, 1, 1, 1], [0, 1, 1, 0], [0, .5, 1, 0]),
                       ([0, 0, 1, 1, 0, 1, 1, 1], [0, 1, 1, 0], [0, 1.5, 1, 0])):
        bad = Planned(lib, [seg], offs, terms, w, 11, plans=_plans(**base, leaf_group=(lg, "<u4"),
                                                                 group_plan=(gp, "<i4"), group_tie=(gt, "<f4")))
        assert not bad.h and bad.code == -1, (lg, bad.err)


def test_minimum_should_match_in_the_plan_word(lib):
    """slg_score_plans::q_min_match > 1: the sub-query runs as a plan (Sum of leaves at least), the count asked
    for rides in bits 8.. of RoundQuery::plan, the threshold seed is off (its champions are single postings,
    which the matcher may reject); such a batch must fit the few-term kernel's plan instantiation."""
    rng = np.random.default_rng(12)
    seg = random_segment(rng, 4000, 40, 15)
    offs, terms, w = random_queries(rng, 3, 4, 40)
    pl = _plans(q_min_match=([2, 1, 3], "<u4"))
    p = Planned(lib, [seg], offs, terms, w, 11, plans=pl)
    assert p.h, p.err
    sqs, _ = check_structure(p, 11)
    assert p.facts.plan_batch and p.facts.uniform and not p.facts.nested
    assert [int(x) & 0xFF for x in sqs["plan"]] == [1, 0, 1]
    assert [int(x) >> 8 for x in sqs["plan"]] == [2, 0, 3]
    assert float(sqs[0]["theta0"]) == 0.0 and float(sqs[2]["theta0"]) == 0.0
    p.close()
    # DisMax root keeps its kind
    pd = Planned(lib, [seg], offs, terms, w, 11,
                 plans=_plans(q_min_match=([2, 2, 2], "<u4"), q_plan=([1, 1, 1], "<i4"), q_tie=([0.5] * 3, "<f4")))
    assert pd.h and [int(x) & 0xFF for x in pd.array(0, RQ)["plan"]] == [2, 2, 2]
    pd.close()
    # nine lists: no kernel counts leaves there
    o9 = np.array([0, 9], dtype=np.uint32)
    bad = Planned(lib, [seg], o9, np.arange(9, dtype=np.uint32), np.ones(9, np.float32), 11,
                  plans=_plans(q_min_match=([2], "<u4")))
    assert not bad.h and bad.code == -4
    ok = Planned(lib, [seg], o9, np.arange(9, dtype=np.uint32), np.ones(9, np.float32), 11,
                 plans=_plans(q_min_match=([1], "<u4")))
    assert ok.h
    ok.close()


def test_score_trees_given_node_by_node(lib):
    """slg_score_plans::q_node_offsets: trees of one and two levels resolve into the root / group forms;
    deeper ones into the canonical node table (every leaf at the same depth, a chain of one-child Sum
    nodes under a leaf that hangs higher up); malformed trees are rejected."""
    rng = np.random.default_rng(12)
    seg = random_segment(rng, 2000, 40, 15)
    offs, terms, w = random_queries(rng, 3, 6, 40)
    leaf = np.tile(np.array([0, 0, 1, 2, 2, 3], dtype=np.uint32), 3)  # 4 leaves per query
    S, D, L = 0, 1, 2
    trees = [  # query 0: one level; query 1: two levels; query 2: three levels with a leaf at every level
        ([D, L, L, L, L], [.3, 0, 0, 0, 0], [0, 0, 0, 0, 0]),
        ([S, D, L, L, L, D, L], [0, .5, 0, 0, 0, 1.0, 0], [0, 0, 1, 1, 0, 0, 5]),
        ([S, L, D, L, S, L, L], [0, 0, .25, 0, 0, 0, 0], [0, 0, 0, 2, 2, 4, 4]),
    ]
    qno = np.cumsum([0] + [len(t[0]) for t in trees])
    pl = _plans(q_leaf=(leaf, "<u4"), q_node_offsets=(qno, "<u4"), node_kind=(sum((t[0] for t in trees), []), "<i4"),
                node_tie=(sum((t[1] for t in trees), []), "<f4"), node_parent=(sum((t[2] for t in trees), []), "<u4"))
    p = Planned(lib, [seg], offs, terms, w, 11, plans=pl)
    assert p.h, p.err
    sqs, tr = check_structure(p, 11)
    assert p.facts.plan_batch and p.facts.nested and p.facts.deep and not p.facts.uniform
    assert [int(x) for x in sqs["plan"]] == [2, 1, 1] and abs(float(sqs["tie"][0]) - 0.3) < 1e-7
    assert [int(x) for x in sqs["n_groups"]] == [0, 3, 0] and [int(x) for x in sqs["depth"]] == [0, 0, 3]
    # query 1 as groups: leaves 0,1 -> DisMax group 0 (tie .5), leaf 2 -> bare leaf = Sum group 1, leaf 3 -> DisMax group 2
    t1 = tr[int(sqs[1]["term_begin"]):int(sqs[1]["term_begin"]) + 6]
    assert [int(g) & 0xFF for g in t1["gmeta"]] == [0, 0, 0, 1, 1, 2]
    assert [(int(g) >> 16) & 1 for g in t1["gmeta"]] == [1, 1, 1, 0, 0, 1]
    # query 2, canonical nodes: 0 root Sum (2 children) | 1,2 pads under leaf 0 (levels 1, 2) | 3 DisMax (2 children)
    # | 4 pad under leaf 1 (level 2) | 5 Sum (2 children, level 2)
    nodes = p.array(9, PN)
    nb = int(sqs[2]["node_begin"])
    got = [(int(n["parent"]), int(n["n_children"]), int(n["kind"])) for n in nodes[nb:nb + 6]]
    assert got == [(0, 2, 0), (0, 1, 0), (1, 1, 0), (0, 2, 1), (3, 1, 0), (3, 2, 0)], got
    t2 = tr[int(sqs[2]["term_begin"]):int(sqs[2]["term_begin"]) + 6]
    assert [int(g) for g in t2["gmeta"]] == [2, 2, 4, 5, 5, 5]   # the level-2 node every list's leaf hangs off
    p.close()
    # malformed: a parent that comes later, a childless Sum, a tie outside [0, 1], too deep, a leaf with a child
    for kind, tie, parent in (([S, L, L, L, L], [0] * 5, [0, 2, 0, 0, 0]),
                              ([S, L, L, L, L, S], [0] * 6, [0, 0, 0, 0, 0, 0]),
                              ([D, L, L, L, L], [1.5, 0, 0, 0, 0], [0] * 5),
                              ([S, S, S, S, S, L, L, L, L], [0] * 9, [0, 0, 1, 2, 3, 4, 4, 4, 4]),
                              ([S, L, L, L, L], [0] * 5, [0, 0, 1, 0, 0])):
        bad = _plans(q_leaf=(leaf[:6], "<u4"), q_node_offsets=([0, len(kind)], "<u4"), node_kind=(kind, "<i4"),
                     node_tie=(tie, "<f4"), node_parent=(parent, "<u4"))
        b = Planned(lib, [seg], offs[:2], terms[:6], w[:6], 11, plans=bad)
        assert not b.h and b.code in (-1, -4), (kind, b.err)


def test_large_batches_plan_identically_on_several_threads(lib):
    """>= 8192 sub-queries are planned by several threads and stitched in query order: the result
    must not depend on the split (compare with the same queries planned in two halves)."""
    rng = np.random.default_rng(11)
    segs = [random_segment(rng, 800, 30, 10) for _ in range(4)]
    champs = [champions_of(s) for s in segs]
    nq = 2304
    offs, terms, w = random_queries(rng, nq, 3, 30, n_segs=4)
    whole = Planned(lib, segs, offs, terms, w, 11, champs=champs)
    assert whole.h and whole.facts.n_sq >= 8192
    sq_w, tr_w = check_structure(whole, 11)
    h = nq // 2
    a = Planned(lib, segs, offs[:h + 1], terms[:3 * h], w[:3 * h], 11, champs=champs)
    b = Planned(lib, segs, offs[h:] - offs[h], terms[3 * h:], w[3 * h:], 11, champs=champs)
    sq_a, sq_b = a.array(0, RQ), b.array(0, RQ)
    assert len(sq_w) == len(sq_a) + len(sq_b)
    for name in ("seg", "n_terms", "n_rounds", "rounds_per_slice", "n_slices", "longest", "ess_mask", "theta0"):
        assert np.array_equal(sq_w[name], np.concatenate([sq_a[name], sq_b[name]])), name
    assert np.array_equal(sq_w["q"], np.concatenate([sq_a["q"], sq_b["q"] + h]))
    assert tr_w.tobytes() == a.array(1, TR).tobytes() + b.array(1, TR).tobytes()
    for p in (whole, a, b):
        p.close()
