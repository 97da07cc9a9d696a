"""Multi-GPU layer: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI).

The path shards in two ways:
  * index sharding  — rank r holds shard r of the index (a searchlite segment, with its own
    docs/df/avgdl as the reference scores each segment independently,
    api/reader.rs:2985-2995); every rank scores ALL queries against its shard, the per-shard
    top-k (Q*k*{doc,seg,score} + Q counts) are exchanged with ONE all-gather, and every rank
    merges them by (score desc, shard asc, segment asc, doc asc) — api/reader.rs:2776-2778,
    query/sort.rs:80-93 with segment_ord = shard*seg_stride + seg;
  * query sharding  — replicas of the index, each rank scores its own queries; the all-gather
    just concatenates results.
No collective sits inside the scoring path itself.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np


class _DevArray:
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr,
                                         "data": (int(ptr), False), "version": 2}


def batch_result_tensors(batch):
    """Zero-copy torch views of a PreparedBatch's device result arrays."""
    import torch
    d_doc, d_seg, d_score, d_count = batch.device_results()
    nq, k = batch.nq, batch.k
    return (torch.as_tensor(_DevArray(d_doc, (nq, k), "<i4"), device="cuda"),
            torch.as_tensor(_DevArray(d_seg, (nq, k), "<i4"), device="cuda"),
            torch.as_tensor(_DevArray(d_score, (nq, k), "<f4"), device="cuda"),
            torch.as_tensor(_DevArray(d_count, (nq,), "<i4"), device="cuda"))


def batch_result_block(batch):
    """Zero-copy int32 view of the contiguous doc|seg|score|count block of a PreparedBatch."""
    import torch
    ptr, nb = batch.device_result_block()
    return torch.as_tensor(_DevArray(ptr, (nb // 4,), "<i4"), device="cuda")


def split_result_block(block, nq: int, k: int):
    """[..., (3k+1)*nq] int32 block(s) -> (doc, seg, score(f32 view), count) views."""
    import torch
    n = nq * k
    lead = tuple(block.shape[:-1])
    doc = block[..., :n].reshape(lead + (nq, k))
    seg = block[..., n:2 * n].reshape(lead + (nq, k))
    score = block[..., 2 * n:3 * n].view(torch.float32).reshape(lead + (nq, k))
    count = block[..., 3 * n:3 * n + nq]
    return doc, seg, score, count


def all_gather_block(block, group=None):
    """ONE collective per step: every rank's result block -> [world, (3k+1)*nq]."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    out = torch.empty((world * block.shape[0],), dtype=block.dtype, device=block.device)
    dist.all_gather_into_tensor(out, block, group=group)
    return out.view(world, block.shape[0])


def all_gather_topk(doc, seg, score, count, group=None):
    """One logical exchange of the per-rank top-k: -> tensors shaped [world, nq, k] / [world, nq]."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)

    def gather(x):
        x = x.contiguous()
        # concatenated along dim 0 (accepted by both the nccl/RCCL and gloo backends), then viewed
        out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x, group=group)
        return out.view((world,) + tuple(x.shape))

    g_doc, g_seg, g_score, g_count = gather(doc), gather(seg), gather(score), gather(count)
    return g_doc, g_seg, g_score, g_count


def merge_shards_host(g_doc, g_seg, g_score, g_count, k: int, seg_stride: int = 1):
    """Host mirror of the cross-segment merge (api/reader.rs:2776-2778) for host-resident
    per-shard results: order by (score desc [f32 total order], shard*seg_stride+seg asc, doc asc).
    Used where the gathered results already live on the host (gloo); device-resident results
    go through slg_merge_shards_device."""
    g_doc = np.asarray(g_doc).astype(np.int64) & 0xFFFFFFFF
    g_seg = np.asarray(g_seg).astype(np.int64) & 0xFFFFFFFF
    g_score = np.asarray(g_score, dtype=np.float32)
    g_count = np.asarray(g_count).astype(np.int64)
    world, nq, _ = g_doc.shape
    out_doc = np.zeros((nq, k), dtype=np.uint32)
    out_seg = np.zeros((nq, k), dtype=np.uint32)
    out_score = np.zeros((nq, k), dtype=np.float32)
    out_count = np.zeros(nq, dtype=np.uint32)
    for q in range(nq):
        rows = []
        for sh in range(world):
            n = int(g_count[sh, q])
            bits = g_score[sh, q, :n].view(np.int32).astype(np.int64)
            key = np.where(bits < 0, bits ^ 0x7FFFFFFF, bits)  # f32::total_cmp order
            for i in range(n):
                rows.append((-int(key[i]), sh * seg_stride + int(g_seg[sh, q, i]),
                             int(g_doc[sh, q, i]), float(g_score[sh, q, i])))
        rows.sort(key=lambda r: (r[0], r[1], r[2]))
        rows = rows[:k]
        out_count[q] = len(rows)
        for i, r in enumerate(rows):
            out_seg[q, i], out_doc[q, i], out_score[q, i] = r[1], r[2], np.float32(r[3])
    return out_doc, out_seg, out_score, out_count


class ShardedSearcher:
    """Index-sharded search: this rank's GpuIndex is shard `rank` of the logical index."""

    def __init__(self, index, seg_stride: Optional[int] = None, group=None):
        import torch
        import torch.distributed as dist
        self.index = index
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.seg_stride = seg_stride if seg_stride is not None else index.n_segs
        index.set_stream(torch.cuda.current_stream().cuda_stream)

    def prepare(self, q_offsets, q_terms, q_weights, k: int, strategy=None):
        from . import searcher
        return self.index.prepare(q_offsets, q_terms, q_weights, k,
                                  searcher.Wand if strategy is None else strategy)

    def run(self, batch) -> Tuple["object", "object", "object", "object"]:
        """score locally -> all-gather -> device merge.  Returns device tensors
        (doc[nq,k], seg[nq,k] = shard*seg_stride+seg, score[nq,k], count[nq])."""
        import torch
        batch.run()
        nq, k = batch.nq, batch.k
        g = all_gather_block(batch_result_block(batch), group=self.group)  # one exchange
        g_doc, g_seg, g_score, g_count = split_result_block(g, nq, k)
        g_doc, g_seg, g_score, g_count = (g_doc.contiguous(), g_seg.contiguous(),
                                          g_score.contiguous(), g_count.contiguous())
        m_doc = torch.empty((nq, k), dtype=torch.int32, device="cuda")
        m_seg = torch.empty_like(m_doc)
        m_score = torch.empty((nq, k), dtype=torch.float32, device="cuda")
        m_count = torch.empty((nq,), dtype=torch.int32, device="cuda")
        self.index.merge_shards_device(self.world, nq, k, g_doc.data_ptr(), g_seg.data_ptr(),
                                       g_score.data_ptr(), g_count.data_ptr(), self.seg_stride,
                                       m_doc.data_ptr(), m_seg.data_ptr(), m_score.data_ptr(),
                                       m_count.data_ptr())
        return m_doc, m_seg, m_score, m_count
