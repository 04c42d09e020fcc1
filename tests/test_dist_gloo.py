"""N > 1 path on CPU: world_size-2 gloo run of the segment-sharded search plumbing (rucene_amd/dist.py) — shard
placement, doc_base, shard-0 statistics, the all-gather layout and the canonical merge. The per-shard search here is
the CPU oracle (tests may use it) and the merge is a numpy restatement of k_merge_lists' contract; on GPUs the same
plumbing runs with rgpu_search_batch_sharded (bench.py --gpus N).

The PRODUCT's half of the same contract is pinned on the GPU side (-m gpu, tests/test_gpu_parity.py):
  * test_shard_records_merge_like_finish_parallel — two shards with different doc_base are searched on one device, each
    into its slot of what would be the all-gather's receive buffer (rgpu_search_batch_record_device), and
    rgpu_merge_records_device (the very k_merge_lists launch, same strides) must give the two-leaf oracle's rows;
  * test_sharded_search_through_the_c_abi_with_a_world_of_one[forced_all_gather] — the in-place ncclAllGather, the cross-stream
    ordering between consecutive collectives and rgpu_comm_reserve EXECUTE (rgpu_config.comm_force_gather), counted by
    rgpu_comm_gathers_issued;
  * test_a_failing_shard_still_joins_the_collective — a failing local search leaves its status in the record and joins."""
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


def _numpy_merge(hits_all, totals_all):
    """[world, nq, k] packed hits -> canonical top-k (score desc, doc asc), counts summed."""
    import torch
    h = hits_all.numpy().view(np.dtype([("doc", "<i4"), ("score", "<f4")]))
    world, nq, k = h.shape[0], h.shape[1], h.shape[2]
    out = np.zeros((nq, k), dtype=h.dtype)
    out["doc"] = -1
    for q in range(nq):
        cand = h[:, q, :].reshape(-1)
        cand = cand[cand["doc"] >= 0]
        order = np.lexsort((cand["doc"], -cand["score"].astype(np.float64)))[:k]
        out[q, :order.size] = cand[order]
    return torch.from_numpy(out.view(np.int64).reshape(nq, k)), totals_all.sum(dim=0)


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from oracle import binding as orc
    from rucene_amd import dist as rdist
    from rucene_amd import indexgen
    dist.init_process_group("gloo", rank=rank, world_size=world)
    docs_per_shard, vocab, k = 60_000, 4_000, 10
    seg = indexgen.build_zipf(docs_per_shard, vocab, shard=rank, doc_base=rank * docs_per_shard)
    shard0 = seg if rank == 0 else indexgen.build_zipf(docs_per_shard, vocab, shard=0)
    specs = [(orc.OP_TERM, [t]) for t in (0, 7, 300, 3_999)] + [(orc.OP_AND, [0, 3, 9]), (orc.OP_OR, [2, 40, 900])]
    # every shard scores with shard 0's statistics (searcher.rs:311-351: the first of the equal-sized largest
    # leaves) — shipped by the host, exactly what bench.py does for the GPU searcher
    stats_leaf = orc.Segment(shard0.doc_bytes, shard0.norms, shard0.max_doc, shard0.terms, doc_base=0,
                             sum_total_term_freq=shard0.sum_total_term_freq)
    mine = orc.Segment(seg.doc_bytes, seg.norms, seg.max_doc, seg.terms, doc_base=rank * docs_per_shard,
                       sum_total_term_freq=seg.sum_total_term_freq)
    searcher = orc.Searcher([mine])
    searcher.override_statistics(stats_leaf, docs_per_shard * world)

    def local_search():
        hits = np.zeros((len(specs), k), dtype=np.dtype([("doc", "<i4"), ("score", "<f4")]))
        hits["doc"] = -1
        totals = np.zeros(len(specs), dtype=np.int64)
        for i, (op, tids) in enumerate(specs):
            d, s, total = searcher.search(op, tids, k, tie_mode=orc.TIE_CANONICAL)
            hits["doc"][i, :d.size] = d
            hits["score"][i, :d.size] = s
            totals[i] = total
        return torch.from_numpy(hits.view(np.int64).reshape(len(specs), k)), torch.from_numpy(totals)

    merged_hits, merged_totals = rdist.sharded_search(local_search, _numpy_merge)
    if rank == 0:
        np.save(out_path, np.concatenate([merged_hits.numpy().reshape(-1), merged_totals.numpy().reshape(-1)]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_search_matches_multi_segment_oracle(oracle, tmp_path):
    import torch.multiprocessing as mp
    from rucene_amd import indexgen
    world, port = 2, _free_port()
    out_path = str(tmp_path / "merged.npy")
    mp.spawn(_worker, args=(world, port, out_path), nprocs=world, join=True)
    got = np.load(out_path)
    # reference: one oracle searcher over both shards as two leaves (leaf 0 = shard 0 provides the statistics)
    docs_per_shard, vocab, k = 60_000, 4_000, 10
    segs = [indexgen.build_zipf(docs_per_shard, vocab, shard=r) for r in range(world)]
    osegs = [oracle.Segment(s.doc_bytes, s.norms, s.max_doc, s.terms, doc_base=r * docs_per_shard,
                            sum_total_term_freq=s.sum_total_term_freq) for r, s in enumerate(segs)]
    osearcher = oracle.Searcher(osegs)
    specs = [(oracle.OP_TERM, [t]) for t in (0, 7, 300, 3_999)] + [(oracle.OP_AND, [0, 3, 9]), (oracle.OP_OR, [2, 40, 900])]
    nq = len(specs)
    hits = got[:nq * k].view(np.dtype([("doc", "<i4"), ("score", "<f4")])).reshape(nq, k)
    totals = got[nq * k:]
    for i, (op, tids) in enumerate(specs):
        d, s, total = osearcher.search(op, tids, k, tie_mode=oracle.TIE_CANONICAL)
        assert totals[i] == total
        assert (hits[i]["doc"][:d.size] == d).all(), (i, hits[i]["doc"], d)
        assert (hits[i]["score"][:d.size].view(np.int32) == s.view(np.int32)).all()
