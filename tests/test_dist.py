"""N > 1 path on CPU: world_size-2 gloo processes exchange per-shard top-k with the same
all_gather_topk() the GPU path uses and merge them; the merged result must equal the oracle
run on the two shards as two segments (api/reader.rs:2670-2778).  The local per-shard scorer is
the oracle here (no GPU in this container); the GPU path is covered by tests/test_gpu_*.py and
bench.py --gpus N."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from oracle import oracle as O
    from searchlite_amd import dist as sdist
    from tests.util import random_queries, random_segment
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        segs = [random_segment(np.random.default_rng(500 + r), 600 + 50 * r, 30, 20) for r in range(world)]
        offs, terms, w = random_queries(np.random.default_rng(9), 12, 3, 30, n_segs=world)
        k = 11
        mine = O.search_batch([segs[rank]], offs, terms[:, rank:rank + 1], w, k, strategy=O.BM25)
        t = [torch.from_numpy(mine[0].astype(np.int32)), torch.from_numpy(mine[1].astype(np.int32)),
             torch.from_numpy(mine[2]), torch.from_numpy(mine[3].astype(np.int32))]
        g = sdist.all_gather_topk(*t)
        merged = sdist.merge_shards_host(*[x.numpy() for x in g], k=k, seg_stride=1)
        want = O.search_batch(segs, offs, terms, w, k, strategy=O.BM25)
        ok = all(np.array_equal(a, b) for a, b in zip(merged[:2], want[:2])) and \
            np.array_equal(merged[2].view(np.uint32), want[2].view(np.uint32)) and \
            np.array_equal(merged[3], want[3])
        open(os.path.join(out_dir, f"rank{rank}.txt"), "w").write("ok" if ok else "mismatch")
    finally:
        dist.destroy_process_group()


def test_index_sharded_merge_gloo_world2(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert open(tmp_path / f"rank{r}.txt").read() == "ok"


def test_merge_shards_host_ordering():
    from searchlite_amd.dist import merge_shards_host
    # two shards, equal scores: shard (segment_ord) then doc decide; negative zero sorts below zero
    g_doc = np.array([[[5, 9, 0]], [[1, 2, 0]]], dtype=np.int32)
    g_seg = np.zeros_like(g_doc)
    g_score = np.array([[[2.0, 1.0, 0.0]], [[2.0, 1.0, 0.0]]], dtype=np.float32)
    g_count = np.array([[2], [2]], dtype=np.int32)
    d, s, sc, c = merge_shards_host(g_doc, g_seg, g_score, g_count, k=3, seg_stride=1)
    assert c[0] == 3
    assert list(zip(s[0], d[0])) == [(0, 5), (1, 1), (0, 9)]
