"""A6 study (CPU, numpy): what block-level skipping could save on the config-3 shape.

For sample queries (5 terms, ranks uniform in [64, 8192), Zipf corpus with config 3's vocabulary;
list densities do not depend on N, so a 2M-doc corpus stands in for the 10M one) the lists are
classified as the host planner does (theta0 = max_t champ[t][rank(k)], lists in ascending order of
their maximum impact are non-essential while the sum of maxima stays below theta0), then measured:
  * the share of postings in non-essential lists (what MaxScore turns into probes);
  * the share of 128-posting blocks (index/postings.rs:11) of non-essential lists whose doc range
    holds NO doc of an essential list = blocks a block-max / doc-range test could skip;
  * the same for 16-posting pieces (one 64-byte sector of doc ids);
  * the share of non-essential postings whose doc is in an essential list (= impacts needed);
  * with theta raised to the TRUE k-th best score (the best any threshold feedback could reach):
    the same shares.
usage: python tools/blockmax_study.py [n_docs] [n_queries] [k]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from searchlite_amd import corpus  # noqa: E402


def impacts(seg, t):
    d, tf = seg.postings(t)
    df = np.float32(len(d))
    idf = np.float32(max(np.log((np.float32(seg.docs) - df + np.float32(0.5)) / (df + np.float32(0.5))), 0) + 1)
    dl = seg.field_doc_len[0][d]
    avg = seg.field_avgdl[0]
    tf = tf.astype(np.float32)
    denom = tf + np.float32(seg.k1) * (np.float32(1) - np.float32(seg.b) + np.float32(seg.b) * dl / avg)
    return d, idf * (tf * (np.float32(seg.k1) + 1)) / denom


def study(seg, terms, k, theta=None):
    lists = [impacts(seg, int(t)) for t in terms]
    ub = np.array([x.max() for _, x in lists])
    if theta is None:
        theta = max(np.sort(x)[-min(k, len(x))] for _, x in lists)
    order = np.argsort(ub)
    ne, acc = [], 0.0
    for i in order[:-1]:
        acc += ub[i]
        if acc < theta:
            ne.append(i)
        else:
            break
    ess = [i for i in range(len(lists)) if i not in ne]
    ess_docs = np.unique(np.concatenate([lists[i][0] for i in ess]))
    tot = sum(len(d) for d, _ in lists)
    p_ne = sum(len(lists[i][0]) for i in ne)
    blocks = skippable = pieces = piece_skip = hits = 0
    for i in ne:
        d = lists[i][0]
        for size, is_block in ((128, True), (16, False)):
            last = d[np.minimum(np.arange(size - 1, len(d) + size - 1, size), len(d) - 1)]
            first = d[np.arange(0, len(d), size)]
            lo = np.searchsorted(ess_docs, first, side="left")
            hi = np.searchsorted(ess_docs, last, side="right")
            empty = int((hi == lo).sum())
            if is_block:
                blocks += len(first)
                skippable += empty
            else:
                pieces += len(first)
                piece_skip += empty
        hits += int(np.isin(d, ess_docs, assume_unique=True).sum())
    # true k-th best score
    all_docs = np.concatenate([d for d, _ in lists])
    all_imp = np.concatenate([x for _, x in lists])
    u, inv = np.unique(all_docs, return_inverse=True)
    score = np.zeros(len(u), dtype=np.float64)
    np.add.at(score, inv, all_imp)
    kth = float(np.sort(score)[-min(k, len(score))])
    return dict(theta=float(theta), kth=kth, tot=tot, p_ne=p_ne, n_ne=len(ne), blocks=blocks, skippable=skippable,
                pieces=pieces, piece_skip=piece_skip, hits=hits)


def main():
    n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    nq = int(sys.argv[2]) if len(sys.argv) > 2 else 48
    k = int(sys.argv[3]) if len(sys.argv) > 3 else 101
    vocab = 1 << 20
    seg = corpus.zipf_segment(n_docs, vocab, seed=43)
    offs, terms, _ = corpus.zipf_queries(nq, 5, seed=7, vocab=vocab)
    for label, use_true in (("seed threshold theta0 (what the planner has)", False),
                            ("the TRUE k-th best score as threshold (upper limit of any feedback)", True)):
        agg = dict(tot=0, p_ne=0, blocks=0, skippable=0, pieces=0, piece_skip=0, hits=0, n_ne=0)
        for q in range(nq):
            t = terms[offs[q]:offs[q + 1]]
            r = study(seg, t, k)
            if use_true:
                r = study(seg, t, k, theta=r["kth"])
            for key in agg:
                agg[key] += r[key]
        print(f"--- {label}: {nq} queries, {n_docs} docs, k = {k}")
        print(f"non-essential lists per query      {agg['n_ne'] / nq:.2f} of 5")
        print(f"postings in non-essential lists    {agg['p_ne'] / agg['tot'] * 100:.1f} %")
        print(f"128-posting blocks skippable       {agg['skippable'] / max(agg['blocks'], 1) * 100:.2f} %  (no essential doc in the block's doc range)")
        print(f"16-posting pieces skippable        {agg['piece_skip'] / max(agg['pieces'], 1) * 100:.2f} %")
        print(f"non-essential postings that hit    {agg['hits'] / max(agg['p_ne'], 1) * 100:.2f} %  (doc also in an essential list)")


if __name__ == "__main__":
    main()
