"""Randomised parity run (GPU box): random corpora, query shapes, k, weights (also zero / negative),
tombstones, doc filters, score plans and strategies, each batch compared bit for bit with the CPU
oracle.  usage: python tools/fuzz_parity.py [iterations] [seed]
(SLG_MAXSCORE=1 / SLG_UNIFORM_MAX_TERMS=0 in the environment force pruning / the many-term kernel;
FUZZ_MANY_LISTS=1 draws MaxScore-classified queries of 14..32 lists instead; FUZZ_TREES=1 turns the
score plans of the standard cases into random two-level trees, FUZZ_DEEP=1 into random trees of up to
four levels given node by node, FUZZ_MIN_MATCH=1 gives every query <= 8 lists, a flat plan and a
minimum_should_match of 0..3; FUZZ_FEW_LISTS=1 keeps every query at
<= 8 lists without plans, so every batch runs on the few-term kernel.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import searchlite_amd as sa
from oracle import oracle as O
from tests.util import random_segment, random_multifield_segment, skewed_segment, assert_same_hits



def run_case(seed0, it, tuning=None):
      """One random batch, keyed (seed0, it): GPU through the C ABI vs the oracle, bit for bit."""
      rng = np.random.default_rng(seed0 * 100003 + it)
      n_segs = int(rng.integers(1, 4))
      skew = rng.random() < 0.3          # big sparse / clustered lists: window and overflow cuts
      multi = not skew and rng.random() < 0.4
      vocab = int(rng.integers(4, 14)) if skew else int(rng.integers(3, 60))
      F = int(rng.integers(2, 4)) if multi else 1
      segs = []
      for s in range(n_segs):
          if skew:
              segs.append(skewed_segment(rng, int(rng.choice([60_000, 500_000, 3_000_000])), vocab))
              if rng.random() < 0.4:
                  segs[-1].set_deleted(np.nonzero(rng.random(segs[-1].n_docs) < 0.2)[0].tolist())
              continue
          n_docs = int(rng.choice([37, 300, 2500, 12000]))
          avg = int(rng.integers(2, 30))
          sg = random_multifield_segment(rng, n_docs, vocab, F, avg) if multi else \
              random_segment(rng, n_docs, vocab, avg, missing_len_frac=0.05 if rng.random() < 0.3 else 0.0,
                             zipf=rng.random() < 0.7)
          if rng.random() < 0.5:
              sg.set_deleted(np.nonzero(rng.random(n_docs) < rng.choice([0.02, 0.3, 0.9]))[0].tolist())
          segs.append(sg)
      nq = int(rng.integers(1, 24))
      k = int(rng.choice([1, 2, 11, 64, 65, 101, 256, 257, 600, 1024, 1025, 3000]))
      MM = os.environ.get("FUZZ_MIN_MATCH", "0") != "0"  # <= 8 lists, flat plans, minimum_should_match 0..3 per query
      FEW = os.environ.get("FUZZ_FEW_LISTS", "0") != "0" or MM
      frng = np.random.default_rng(seed0 * 1000033 + it + 5)
      offs, terms, w, leaf, plan, tie, nl = [0], [], [], [], [], [], []
      V = vocab * F
      for q in range(nq):
          T = int(rng.choice([0, 1, 2, 3, 4, 5, 6, 8, 13, 32], p=[.03, .1, .1, .2, .12, .12, .1, .1, .08, .05]))
          if FEW:  # every batch on the few-term kernel: <= 8 lists, no plans (drawn from a generator of its own)
              T = int(frng.choice([0, 1, 2, 3, 4, 5, 6, 7, 8], p=[.03, .07, .1, .2, .1, .15, .1, .1, .15]))
          T = min(T, V)
          ids = rng.choice(V, size=T, replace=False)
          lf = np.sort(rng.integers(0, max(1, T // 2 + 1), size=T)) if rng.random() < 0.5 else np.arange(T)
          if rng.random() < 0.3:
              lf = rng.permutation(lf)      # leaves need not arrive sorted
          for i in range(T):
              row = []
              for s in range(n_segs):
                  row.append(sa.NO_TERM if rng.random() < 0.1 else int(ids[i]))
              terms.append(row)
              r = rng.random()
              w.append(np.float32(0.0 if r < 0.03 else (-rng.random() if r < 0.08 else rng.random() * 3 + 0.1)))
              leaf.append(int(lf[i]))
          offs.append(len(terms))
          dm = rng.random() < 0.35
          plan.append(sa.PLAN_DISMAX if dm else sa.PLAN_SUM)
          tie.append(float(rng.choice([0.0, 0.3, 1.0])) if dm else 0.0)
          nl.append(int(max(lf) + 1 + rng.integers(0, 2)) if T else int(rng.integers(0, 2)))
      offs = np.array(offs, dtype=np.uint32)
      terms = np.array(terms, dtype=np.uint32).reshape(-1, n_segs)
      w = np.array(w, dtype=np.float32)
      use_plan = (rng.random() < 0.6 and not FEW) or MM
      kw = dict(q_leaf=np.array(leaf, dtype=np.uint32), q_plan=np.array(plan, dtype=np.int32),
                q_tie=np.array(tie, dtype=np.float32), q_nleaves=np.array(nl, dtype=np.uint32)) if use_plan else {}
      if use_plan and os.environ.get("FUZZ_TREES", "0") != "0":
          # two-level plans: the leaves of every query cut into consecutive groups, each Sum or DisMax
          # (drawn from a generator of its own, so the keyed flat cases stay what they were)
          trng = np.random.default_rng(seed0 * 1000003 + it + 77)
          qlo, lg, qgo, gp, gt = [0], [], [0], [], []
          for q in range(nq):
              g = 0
              for l in range(nl[q]):
                  if l and trng.random() < 0.5:
                      g += 1
                  lg.append(g)
              ng = g + 1 if nl[q] else 0
              for _ in range(ng):
                  dm = trng.random() < 0.5
                  gp.append(sa.PLAN_DISMAX if dm else sa.PLAN_SUM)
                  gt.append(float(trng.choice([0.0, 0.5, 1.0])) if dm else 0.0)
              qlo.append(len(lg))
              qgo.append(len(gp))
          kw.update(q_leaf_offsets=np.array(qlo, dtype=np.uint32), leaf_group=np.array(lg, dtype=np.uint32),
                    q_group_offsets=np.array(qgo, dtype=np.uint32), group_plan=np.array(gp, dtype=np.int32),
                    group_tie=np.array(gt, dtype=np.float32))
      if use_plan and os.environ.get("FUZZ_DEEP", "0") != "0":
          # trees of up to four levels of Sum / DisMax given node by node (pre-order; the leaves in leaf order)
          trng = np.random.default_rng(seed0 * 1000003 + it + 177)
          qno, nk, nt, npar = [0], [], [], []
          for q in range(nq):
              base = len(nk)

              def node(par, n_leaves, depth):
                  me = len(nk) - base
                  dm = trng.random() < 0.5
                  nk.append(sa.PLAN_DISMAX if dm else sa.PLAN_SUM)
                  nt.append(float(trng.choice([0.0, 0.5, 1.0])) if dm else 0.0)
                  npar.append(par)
                  i = 0
                  while i < n_leaves:
                      take = int(trng.integers(1, n_leaves - i + 1))
                      if depth < 4 and trng.random() < 0.6:
                          node(me, take, depth + 1)
                      else:
                          for _ in range(take):
                              nk.append(2)  # SLG_PLAN_LEAF
                              nt.append(0.0)
                              npar.append(me)
                      i += take
              nl[q] = max(int(nl[q]), 1)  # (a tree has at least one leaf; one without terms scores 0.0)
              node(0, int(nl[q]), 1)
              qno.append(len(nk))
          kw["q_nleaves"] = np.array(nl, dtype=np.uint32)
          kw.update(q_node_offsets=np.array(qno, dtype=np.uint32), node_kind=np.array(nk, dtype=np.int32),
                    node_tie=np.array(nt, dtype=np.float32), node_parent=np.array(npar, dtype=np.uint32))
      mm = None
      if MM:
          mrng = np.random.default_rng(seed0 * 1000003 + it + 277)
          mm = mrng.integers(0, 4, size=nq).astype(np.uint32)
      use_filter = rng.random() < 0.4
      if os.environ.get("FUZZ_NO_FILTER", "0") != "0":  # (debugging a keyed case: same draws, no filter)
          use_filter = False
      masks = [rng.random(sg.n_docs) < rng.choice([0.05, 0.5, 0.95]) for sg in segs]
      with sa.GpuIndex(segs, tuning=tuning) as ix:
          qf = None
          if use_filter:
              fid = ix.add_filter(masks)
              qf = np.array([fid if rng.random() < 0.6 else -1 for _ in range(nq)], dtype=np.int32)
          strat = int(rng.choice([sa.Bm25, sa.Wand, sa.Bmw]))
          got = ix.search_plan(offs, terms, w, k, strategy=strat, q_filter=qf,
                               **(dict(kw, q_min_match=mm) if MM else kw))
      if MM:
          want = O.search_batch_min_match(segs, offs, terms, w, k, mm, strategy=O.BM25,
                                          q_filter=np.where(qf >= 0, 0, -1) if use_filter else None,
                                          filters=[masks] if use_filter else None, **kw)
      elif use_filter:
          want = O.search_batch_filtered(segs, offs, terms, w, k, np.where(qf >= 0, 0, -1), [masks],
                                         strategy=O.BM25, **kw)
      else:
          want = O.search_batch(segs, offs, terms, w, k, strategy=O.BM25, **kw)
      try:
          assert_same_hits(got, want, 0.0, f"fuzz it={it} seed={seed0} nq={nq} k={k} segs={n_segs} multi={multi} "
                                           f"plan={use_plan} filter={use_filter}")
      except AssertionError:
          for q in range(nq):
              n = int(want[3][q])
              if int(got[3][q]) == n and np.array_equal(got[0][q, :n], want[0][q, :n]) and \
                      np.array_equal(got[1][q, :n], want[1][q, :n]):
                  continue
              T = int(offs[q + 1] - offs[q])
              i = 0
              while i < min(n, int(got[3][q])) and got[0][q, i] == want[0][q, i] and got[1][q, i] == want[1][q, i]:
                  i += 1
              gs, gd = int(got[1][q, i]), int(got[0][q, i])
              dead = segs[gs].deleted is not None and bool(np.unpackbits(segs[gs].deleted, bitorder="little")[gd])
              print(f"query {q}: T={T} strat={strat} filter={None if qf is None else int(qf[q])} counts gpu/oracle "
                    f"{int(got[3][q])}/{n} first diff at {i}: gpu (seg {gs}, doc {gd}, {got[2][q, i]}) oracle "
                    f"(seg {int(want[1][q, i])}, doc {int(want[0][q, i])}, {want[2][q, i]}); gpu doc deleted={dead} "
                    f"mask={bool(masks[gs][gd])} plan={plan[q]} tie={tie[q]} nl={nl[q]} "
                    f"leaves={leaf[int(offs[q]):int(offs[q + 1])]} w={w[int(offs[q]):int(offs[q + 1])]} "
                    f"terms={terms[int(offs[q]):int(offs[q + 1])].tolist()}")
              if "q_node_offsets" in kw:
                  a, b_ = int(kw["q_node_offsets"][q]), int(kw["q_node_offsets"][q + 1])
                  print("  nodes kind", kw["node_kind"][a:b_].tolist(), "tie", kw["node_tie"][a:b_].tolist(),
                        "parent", kw["node_parent"][a:b_].tolist())
              break
          raise


def run_case_many_lists(seed0, it, tuning=None):
    """MaxScore-classified queries of 14..32 lists (the chunk's slot descriptors must
    fit the wave's 64 lanes): skewed corpora (clustered and spread lists), one or two heavily
    weighted lists and many lightly weighted dense ones, strategies Wand / Bmw, no filters, so the
    planner classifies most lists non-essential.  Keyed (seed0, it), bit for bit vs the oracle."""
    rng = np.random.default_rng(seed0 * 7919 + it + 0x5EED)
    vocab = int(rng.integers(34, 48))
    n_docs = int(rng.choice([60_000, 500_000, 3_000_000]))
    seg = skewed_segment(rng, n_docs, vocab)
    if rng.random() < 0.3:
        seg.set_deleted(np.nonzero(rng.random(n_docs) < 0.1)[0].tolist())
    nq = int(rng.integers(1, 6))
    k = int(rng.choice([1, 5, 11, 64, 101, 256]))
    offs, terms, w = [0], [], []
    for q in range(nq):
        T = int(rng.integers(14, 33))
        ids = rng.choice(vocab, size=T, replace=False)
        heavy = set(rng.choice(T, size=int(rng.integers(1, 3)), replace=False).tolist())
        for i in range(T):
            terms.append([int(ids[i])])
            w.append(np.float32(rng.random() * 40 + 10) if i in heavy else np.float32(rng.random() * 0.05))
        offs.append(len(terms))
    offs = np.array(offs, dtype=np.uint32)
    terms = np.array(terms, dtype=np.uint32).reshape(-1, 1)
    w = np.array(w, dtype=np.float32)
    with sa.GpuIndex([seg], tuning=tuning) as ix:
        strat = int(rng.choice([sa.Wand, sa.Bmw]))
        got = ix.search_batch(offs, terms, w, k, strat)
    want = O.search_batch([seg], offs, terms, w, k, strategy=O.BM25)
    assert_same_hits(got, want, 0.0, f"fuzz many-lists it={it} seed={seed0} nq={nq} k={k} n_docs={n_docs}")


def run(iters, seed0, verbose=True, tuning=None):
    O.build()
    t0 = time.time()
    many = os.environ.get("FUZZ_MANY_LISTS", "0") != "0"
    for it in range(iters):
        (run_case_many_lists if many else run_case)(seed0, it, tuning)
        if verbose and it % 10 == 9:
            print(f"{it + 1} batches ok ({time.time() - t0:.0f} s)", flush=True)
    if verbose:
        print("fuzz_parity: all", iters, "batches bit-exact")


if __name__ == "__main__":
    if len(sys.argv) > 3:  # one keyed case: fuzz_parity.py 1 <seed> <iteration>
        run_case(int(sys.argv[2]), int(sys.argv[3]), None)
        print("case ok")
        sys.exit(0)
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
