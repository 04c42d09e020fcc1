"""Segment-sharded search across ranks (one process per GPU): the GPU counterpart of per-leaf collectors merged by
TopDocsCollector::finish_parallel (search/collector/top_docs.rs:157-172), with the mpsc channel replaced by ONE
all-gather of per-shard top-k over RCCL/xGMI.

The collective itself lives behind the C ABI (include/rucene_gpu.h: rgpu_comm_*, rgpu_search_batch_sharded — what a
Rust host binds); this module only creates the communicator for a torch.distributed job (the 128-byte RCCL id travels
over the job's own process group) and, for the CPU tests, restates the record layout of that all-gather with gloo.

Each rank evaluates the replicated query batch against its own shard (hits already in global doc ids: doc +
doc_base) and contributes one record: `n_queries x k` rgpu_hit entries, `n_queries` int64 hit counts and one int64 status word.
The payload is tiny (1024 queries x k=10 -> 88 KiB per rank), i.e. latency-bound: batch many queries per collective.
"""
import numpy as np
import torch
import torch.distributed as dist


def record_words(n_queries, k):
    """int64 words of one rank's record: [n_queries x k packed hits][n_queries hit counts][status] (rgpu_record_bytes / 8)."""
    return n_queries * k + n_queries + 1


def pack_record(hits_local, totals_local, status=0):
    """hits_local [n_queries, k] int64 (packed {i32 doc, f32 score}), totals_local [n_queries] int64 -> one record."""
    return torch.cat([hits_local.reshape(-1), totals_local.reshape(-1), torch.tensor([status], dtype=torch.int64)]).contiguous()


def unpack_records(records, world, n_queries, k):
    """[world * record_words] -> (hits_all [world, n_queries, k], totals_all [world, n_queries]) views."""
    r = records.view(world, record_words(n_queries, k))
    return r[:, :n_queries * k].reshape(world, n_queries, k), r[:, n_queries * k:n_queries * k + n_queries]


def all_gather_records(hits_local, totals_local, group=None):
    """The collective of rgpu_search_batch_sharded with torch.distributed (gloo on CPU tensors in the tests): ONE
    all-gather per batch of each rank's record."""
    world = dist.get_world_size(group)
    nq, k = hits_local.shape
    send = pack_record(hits_local, totals_local)
    recv = torch.empty((world * send.numel(),), dtype=send.dtype, device=send.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    return unpack_records(recv, world, nq, k)


def sharded_search(local_search, merge, group=None):
    """local_search() -> (hits_local, totals_local) tensors for this rank's shard;
    merge(hits_all, totals_all) -> (hits, totals). CPU-test form of rgpu_search_batch_sharded."""
    hits_local, totals_local = local_search()
    hits_all, totals_all = all_gather_records(hits_local, totals_local, group)
    return merge(hits_all, totals_all)


def create_comm(ctx, group=None):
    """rgpu_comm over the ranks of a torch.distributed job: rank 0 draws the RCCL unique id (rgpu_comm_unique_id), the
    job's process group carries its 128 bytes to the others, every rank joins (rgpu_comm_init)."""
    from . import _lib
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    uid = _lib.comm_unique_id() if rank == 0 else np.zeros(128, np.uint8)
    box = [bytes(uid)]
    if world > 1:
        dist.broadcast_object_list(box, src=0, group=group)
    return _lib.Comm(ctx, world, rank, np.frombuffer(box[0], dtype=np.uint8))
