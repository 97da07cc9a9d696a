"""Multi-GPU layer: one process per GPU.  The index-sharded data path (RCCL all-gather + device
merge) lives behind the C ABI (slg_shard_group, include/searchlite_gpu.h); the helpers below are
the host-side mirrors used by tests and by bench.py's query-sharded leg.

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


def exchange_unique_id(rank: int, group=None) -> bytes:
    """Control plane only: rank 0 makes the shard group's 128-byte id (slg_shard_unique_id) and the
    others receive it through torch.distributed's object broadcast (any backend; a file, an
    environment variable or MPI would do as well).  The DATA path never touches torch."""
    import torch.distributed as dist
    from . import searcher
    box = [searcher.shard_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    return box[0]


class ShardedSearcher:
    """Index-sharded search: this rank's GpuIndex is shard `rank` of the logical index.  The
    exchange and the merge run behind the C ABI (slg_shard_group / slg_batch_run_sharded: one
    ncclAllGather of the result blocks + merge_shards_kernel); torch.distributed, if present at all,
    only carried the communicator id."""

    def __init__(self, index, rank: int, world: int, unique_id: bytes, segs_per_rank: Optional[int] = None):
        from . import searcher
        self.index, self.rank, self.world = index, rank, world
        self.group = searcher.ShardGroup(index, rank, world, unique_id, segs_per_rank)
        self.seg_stride = self.group.segs_per_rank

    def close(self) -> None:
        self.group.close()

    def prepare(self, q_offsets, q_terms, q_weights, k: int, strategy=None):
        from . import searcher
        return self.index.prepare(q_offsets, q_terms, q_weights, k,
                                  searcher.Wand if strategy is None else strategy)

    def run(self, batch, fetch: bool = True):
        """score locally -> all-gather -> device merge.  fetch=True: merged host arrays
        (doc[nq,k], seg[nq,k] = shard*seg_stride+seg, score[nq,k], count[nq]); fetch=False: the
        merged block stays on the device (batch.sharded_device_results())."""
        return batch.run_sharded(self.group, fetch=fetch)
