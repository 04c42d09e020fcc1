"""Segment-sharded search across ranks (one process per GPU): the GPU counterpart of per-leaf collectors merged by
TopDocsCollector::finish_parallel (search/collector/top_docs.rs:157-172), with the mpsc channel replaced by ONE
all-gather of per-shard top-k over RCCL/xGMI (backend "nccl") — or gloo on CPU tensors in tests.

Each rank evaluates the replicated query batch against its own shard (hits already in global doc ids: doc +
doc_base) and contributes `n_queries x k` rgpu_hit records (viewed as int64) + `n_queries` hit counts. The payload
is tiny (1024 queries x k=10 -> 80 KiB per rank), i.e. latency-bound: batch many queries per collective.
"""
import torch
import torch.distributed as dist


def all_gather_topk(hits_local, totals_local, group=None):
    """hits_local [n_queries, k] int64 (packed {i32 doc, f32 score}), totals_local [n_queries] int64
    -> (hits_all [world, n_queries, k], totals_all [world, n_queries]) on every rank."""
    world = dist.get_world_size(group)
    nq, k = hits_local.shape
    # rank-major concatenation along dim 0 (the layout both gloo and RCCL accept), viewed as [world, ...]
    hits_all = torch.empty((world * nq, k), dtype=hits_local.dtype, device=hits_local.device)
    totals_all = torch.empty((world * nq,), dtype=totals_local.dtype, device=totals_local.device)
    dist.all_gather_into_tensor(hits_all, hits_local.contiguous(), group=group)
    dist.all_gather_into_tensor(totals_all, totals_local.contiguous(), group=group)
    return hits_all.view(world, nq, k), totals_all.view(world, nq)


def sharded_search(local_search, merge, group=None):
    """local_search() -> (hits_local, totals_local) tensors for this rank's shard;
    merge(hits_all, totals_all) -> (hits, totals) — on GPUs `hip_merge(ctx)` below (k_merge_lists)."""
    hits_local, totals_local = local_search()
    hits_all, totals_all = all_gather_topk(hits_local, totals_local, group)
    return merge(hits_all, totals_all)


def hip_merge(ctx):
    """merge() implemented by rgpu_merge_topk_device: canonical order (score desc, doc asc), hit counts summed."""
    def merge(hits_all, totals_all):
        if not hits_all.is_cuda:
            raise RuntimeError("hip_merge needs device tensors: rucene_amd has no CPU fallback")
        world, nq, k = hits_all.shape
        out_h = torch.empty((nq, k), dtype=torch.int64, device=hits_all.device)
        out_t = torch.empty((nq,), dtype=torch.int64, device=hits_all.device)
        stream = torch.cuda.current_stream().cuda_stream
        if stream:  # a real (non-default) torch stream: the merge is simply enqueued behind the collective
            ctx.merge_topk_device(hits_all.data_ptr(), totals_all.data_ptr(), world, nq, k, out_h.data_ptr(), out_t.data_ptr(), stream)
        else:       # default stream: the ctx has its own, so fence on both sides
            torch.cuda.current_stream().synchronize()
            ctx.merge_topk_device(hits_all.data_ptr(), totals_all.data_ptr(), world, nq, k, out_h.data_ptr(), out_t.data_ptr())
            ctx.synchronize()
        return out_h, out_t
    return merge
