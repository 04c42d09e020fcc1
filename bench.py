#!/usr/bin/env python3
"""bench.py — the hot path on BASELINE.json's metric: queries/sec + postings decoded/sec, BM25, 10M-doc Zipfian.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: 1024 single-term BM25 queries (BASELINE.json configs[1],
SURVEY.md §8(d)) evaluated on the GPU against a 10M-doc Zipfian segment that is already resident in HBM
(decode -> BM25 -> top-10). With N GPUs the index is segment-sharded (one 10M-doc shard per rank, shard = rank),
the query batch is replicated, every rank evaluates it against its shard, per-shard top-k is all-gathered over
RCCL and merged on the GPU — weak scaling: the unit is one query evaluated against one 10M-doc segment.
Rank 0 prints ONE JSON line. At N = 1 the line also carries, under "configs", north_star's other targets measured
in the same run: 3-term AND (configs[2]), 10-term OR top-100 (configs[3]), the block-decode microbenchmark, and an
out-of-Infinity-Cache point (a 100M-doc shard, .doc ~ 670 MB) for block decode and the single-term kernel —
each with its isolated kernel time, algorithmic bytes, roofline fraction, a bounded CPU leg and full-batch parity
against the oracle. `--configs none` skips them.
"""
import argparse
import glob
import json
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
SEED_QUERIES = 0x527563656E65 ^ 0x51  # "Rucene" ^ purpose tag


def term_encoded_bytes(terms, doc_len_end):
    """Encoded postings bytes per term (blocks + VInt tail, no skip data), from the term table alone:
    df > 128 -> skip_offset (the skip data starts right after the postings); otherwise next term's start."""
    start = terms["doc_start_fp"].astype(np.int64)
    nxt = np.empty_like(start)
    nxt[:-1] = start[1:]
    nxt[-1] = doc_len_end
    out = np.where(terms["doc_freq"] > 128, terms["skip_offset"], nxt - start)
    out[terms["doc_freq"] <= 1] = 0
    return out.astype(np.int64)


def profiled_traffic(kernel, tag):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC summary profiles/<tag>_rocprofv3_summary.txt:
    2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes. The factor 2 is MI355X_MICROARCH.md's gfx950 correction (FETCH_SIZE
    tallies the 128-byte requests of wide 16-byte-per-lane streaming reads at 64 bytes — these kernels' row loads);
    WRITE_SIZE is taken as reported. None when no such profile is committed — bench.py never runs rocprof."""
    if " + " in kernel:  # several kernels per batch (OR): their per-launch traffic summed
        parts = [profiled_traffic(x, tag) for x in kernel.split(" + ")]
        if any(x is None for x in parts):
            return None
        return {"bytes": sum(x["bytes"] for x in parts), "source": parts[0]["source"]}
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "%s_rocprofv3_summary.txt" % tag))):
        fetch = write = None
        for line in open(path):
            if kernel not in line:
                continue
            m = re.search(r"FETCH_SIZE=([0-9.e+]+)", line)
            if m:
                fetch = float(m.group(1))
            m = re.search(r"WRITE_SIZE=([0-9.e+]+)", line)
            if m:
                write = float(m.group(1))
        if fetch is not None and write is not None:
            best = {"bytes": (2.0 * fetch + write) * 1024.0, "source": os.path.relpath(path, ROOT)}
    return best


def build_queries(n_queries, kind, seed):
    from rucene_amd import indexgen
    if kind == "term":
        ranks = indexgen.log_uniform_ranks(n_queries, 1, 10_000, seed).reshape(-1, 1)
    elif kind == "and3":
        ranks = indexgen.log_uniform_ranks(3 * n_queries, 1, 1000, seed ^ 0xA3).reshape(-1, 3)
    else:
        ranks = indexgen.log_uniform_ranks(10 * n_queries, 1, 10_000, seed ^ 0x0A).reshape(-1, 10)
    return ranks - 1  # term ids


WORKLOAD_TEXT = {
    "term": "1024 single-term BM25 queries top-10, ranks log-uniform 1..10000 (BASELINE configs[1])",
    "and3": "1024 x 3-term AND top-10, ranks log-uniform 1..1000 (BASELINE configs[2])",
    "or10": "1024 x 10-term OR top-100, ranks log-uniform 1..10000 (BASELINE configs[3])",
}
DOMINANT = {"term": "k_search_term", "and3": "k_search_and", "or10": "k_or_wide"}
PROFILE_TAG = {"term": "r02_term", "and3": "r02_and3", "or10": "r02_or10", "decode": "r02_decode"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--docs", type=int, default=10_000_000)
    ap.add_argument("--vocab", type=int, default=1_000_000)
    ap.add_argument("--queries", type=int, default=1024)
    ap.add_argument("--k", type=int, default=0, help="0 = the workload's own (10; 100 for or10)")
    ap.add_argument("--workload", choices=["term", "and3", "or10"], default="term")
    ap.add_argument("--configs", default="all", help="all | none | comma list of and3,or10,block_decode,out_of_cache (N = 1 only)")
    ap.add_argument("--big-docs", type=int, default=100_000_000, help="size of the out-of-cache shard")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true",
                    help="developer check: take the N > 1 code path (RCCL all-gather + device merge) with a world of one")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N > 1)" % (args.gpus, world))

    import torch
    import torch.distributed as dist
    import rucene_amd
    from rucene_amd import indexgen

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: rucene_amd has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist_mode = world > 1 or args.force_dist
    # RCCL prints a start-up banner on fd 1; the contract is ONE JSON line on stdout, so everything native goes to
    # stderr until that line is written
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)
    if dist_mode:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        # torch.distributed is job plumbing here (communicator id, barriers, max of the time): gloo. The data path's
        # collective is the library's own RCCL all-gather (rgpu_search_batch_sharded).
        dist.init_process_group("gloo", rank=rank, world_size=world)
    cores = os.cpu_count() or 1
    nq = args.queries
    T, B = rucene_amd.TermQuery, rucene_amd.BooleanQuery
    ctx = rucene_amd.Context(device=local_rank, profile_kernels=False)
    from rucene_amd import dist as rdist
    # N > 1: the collective lives behind the C ABI (rgpu_comm_*: RCCL linked into librucene_gpu.so); torch.distributed only
    # carries the 128-byte communicator id, the barriers around the timed region and the max-over-ranks of the time
    comm = rdist.create_comm(ctx) if dist_mode else None

    class Shard:
        """One segment resident in HBM + the searcher over it (statistics of shard 0, the first largest leaf)."""

        def __init__(self, docs, shard, doc_base, n_shards):
            t0 = time.time()
            self.seg = indexgen.build_zipf(docs, args.vocab, shard=shard, doc_base=doc_base)
            self.build_s = time.time() - t0
            self.leaf = rucene_amd.LeafReader.from_synthetic(self.seg)
            self.searcher = rucene_amd.GpuIndexSearcher([self.leaf], ctx=ctx)
            stats_seg = self.seg if shard == 0 else indexgen.build_zipf(docs, args.vocab, shard=0)
            self.searcher.collection_statistics = rucene_amd.CollectionStatistics("body", 0, docs * n_shards, stats_seg.doc_count,
                                                                                  stats_seg.sum_total_term_freq)
            stats_df = stats_seg.terms["doc_freq"].astype(np.int64)
            self.searcher.term_statistics = lambda t: int(stats_df[t])
            self.enc = term_encoded_bytes(self.seg.terms, self.seg.doc_bytes.size - 16)

        def queries(self, kind):
            tids = build_queries(nq, kind, SEED_QUERIES)
            if kind == "term":
                qs = [T(int(t[0])) for t in tids]
            elif kind == "and3":
                qs = [B.build([T(int(x)) for x in t], []) for t in tids]
            else:
                qs = [B.build([], [T(int(x)) for x in t]) for t in tids]
            return tids, qs

        def batch(self, kind, k):
            tids, qs = self.queries(kind)
            packed = self.searcher.pack(qs, self.leaf)
            flat = tids.reshape(-1)
            postings = int(self.seg.terms["doc_freq"][flat].sum())
            # encoded blocks + tails + 1 B norm per scored posting + 8 k B output per query
            algo_bytes = int(self.enc[flat].sum()) + postings + 8 * k * nq
            return tids, qs, packed, postings, algo_bytes

    class Lane:
        def __init__(self, k):
            self.stream = torch.cuda.Stream()
            self.hits = torch.empty((nq, k), dtype=torch.int64, device="cuda")      # rgpu_hit {i32 doc, f32 score}
            self.totals = torch.empty((nq,), dtype=torch.int64, device="cuda")
            self.local_hits = torch.empty((nq, k), dtype=torch.int64, device="cuda")  # this rank's shard alone (parity checks)
            self.local_totals = torch.empty((nq,), dtype=torch.int64, device="cuda")

    def measure(shard, kind, k, steps, warmup, two_streams, with_planning=True):
        """Times `steps` passes of one batch. rgpu_search_batch_device only enqueues (staging copy + kernels): back-to-back
        steps overlap the host-side planning of batch i+1 with the kernels of batch i; the timed region is bracketed by a
        barrier + device-wide synchronize on both sides and runs WITHOUT per-kernel events. N > 1: a step =
        rgpu_search_batch_sharded: search -> ONE RCCL all-gather of the per-shard {top-k, count} records -> device merge,
        enqueued in order on one stream without host syncs.
        Returns wall figures for one stream, for two alternating streams (the small merge / scatter kernels and the tail
        of step i then run under step i+1's search kernel), for one stream with the query planning (pack: term
        resolution, BM25 weights, sim table) redone every step, and the isolated per-kernel durations (HIP events on one
        stream, a separate pass)."""
        tids, qs, packed, postings, algo_bytes = shard.batch(kind, k)
        lanes = [Lane(k), Lane(k)]
        state = {"n": 0}
        merged = {}

        def step(pk, n_lanes):
            lane = lanes[state["n"] % n_lanes]
            state["n"] += 1
            merged["hits"], merged["totals"] = lane.hits, lane.totals
            if dist_mode:
                comm.search_batch_sharded(shard.leaf.segment, pk[0], pk[1], k, lane.hits.data_ptr(), lane.totals.data_ptr(), lane.stream.cuda_stream)
            else:
                shard.leaf.segment.search_batch_device(pk[0], pk[1], k, lane.hits.data_ptr(), lane.totals.data_ptr(), lane.stream.cuda_stream)

        def timed(n_lanes, replan):
            for _ in range(warmup):
                step(packed, n_lanes)
            torch.cuda.synchronize()
            if dist_mode:
                dist.barrier()
            torch.cuda.synchronize()
            t = time.perf_counter()
            op = {"term": rucene_amd._lib.OP_TERM, "and3": rucene_amd._lib.OP_AND}.get(kind, rucene_amd._lib.OP_OR)
            for _ in range(steps):
                if replan == "objects":
                    pk = shard.searcher.pack(qs, shard.leaf)
                elif replan == "array":
                    pk = shard.searcher.pack_uniform(op, tids, shard.leaf)
                else:
                    pk = packed
                step(pk, n_lanes)
            torch.cuda.synchronize()
            if dist_mode:
                dist.barrier()
            torch.cuda.synchronize()
            el = time.perf_counter() - t
            if dist_mode:
                tt = torch.tensor([el], dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                el = float(tt.item())
            return 1e3 * el / steps

        res = {"tids": tids, "postings": postings, "algo_bytes": algo_bytes}
        res["ms_one_stream"] = timed(1, False)
        res["ms_two_streams"] = timed(2, False) if two_streams else None
        res["ms_one_stream_with_planning"] = timed(1, "objects") if with_planning else None
        res["ms_two_streams_with_array_planning"] = timed(2, "array") if with_planning else None
        # isolated kernel durations: the same steps on ONE stream with HIP events around every launch
        ctx.set_profiling(True)
        ctx.kernel_stats_reset()
        for _ in range(steps):
            step(packed, 1)
        torch.cuda.synchronize()
        stats = ctx.kernel_stats()
        res["kernels_ms"] = {n: s["total_ms"] / max(1, s["launches"]) for n, s in stats.items()}  # average per launch
        res["kernels_ms_per_step"] = {n: s["total_ms"] / steps for n, s in stats.items()}
        res["launches_per_step"] = {n: s["launches"] / steps for n, s in stats.items()}
        ctx.set_profiling(False)
        ctx.kernel_stats_reset()
        res["g_hits"] = merged["hits"].cpu().numpy().view(rucene_amd.HIT_DTYPE).reshape(nq, k).copy()
        res["g_totals"] = merged["totals"].cpu().numpy().copy()
        if args.force_dist and world == 1:  # a world of one: the gathered + merged rows must equal the plain local search
            lane = lanes[0]
            shard.leaf.segment.search_batch_device(packed[0], packed[1], k, lane.local_hits.data_ptr(), lane.local_totals.data_ptr(), lane.stream.cuda_stream)
            torch.cuda.synchronize()
            res["force_dist_same"] = bool(torch.equal(merged["hits"], lane.local_hits)) and bool(torch.equal(merged["totals"], lane.local_totals))
        return res

    def cpu_baseline_leg(shard, kind, k, res, budget_s, sample_queries):
        """cpu_baseline: the oracle (C++ restatement of Rucene's CPU IndexSearcher, one query per thread on all host cores): a bounded timing
        leg on `sample_queries` of the batch, and parity of the GPU's rows against it on the FULL batch (canonical tie
        rule). A reported baseline, not the target."""
        from oracle import binding as orc  # the checker: only ever imported here, after the timed regions
        osearcher = orc.Searcher([orc.Segment(shard.seg.doc_bytes, shard.seg.norms, shard.seg.max_doc, shard.seg.terms,
                                              sum_total_term_freq=shard.seg.sum_total_term_freq)])
        tids = res["tids"]
        op = {"term": orc.OP_TERM, "and3": orc.OP_AND, "or10": orc.OP_OR}[kind]
        ns = min(nq, sample_queries)
        ops = np.full(ns, op, np.int32)
        offs = (np.arange(ns + 1) * tids.shape[1]).astype(np.int32)
        flat = np.ascontiguousarray(tids[:ns]).reshape(-1)
        spent, done_q, reps = 0.0, 0, 0
        while spent < budget_s and reps < 50:
            _, _, _, _, _, secs = osearcher.search_batch(ops, offs, flat, k, tie_mode=orc.TIE_RUST_HEAP, threads=cores)
            spent += secs
            done_q += ns
            reps += 1
        sample_postings = int(shard.seg.terms["doc_freq"][flat].sum())
        ops = np.full(nq, op, np.int32)
        offs = (np.arange(nq + 1) * tids.shape[1]).astype(np.int32)
        cd, cs, _, ct, _, _ = osearcher.search_batch(ops, offs, tids.reshape(-1), k, tie_mode=orc.TIE_CANONICAL, threads=cores)
        g_hits, g_totals = res["g_hits"], res["g_totals"]
        if kind == "or10":  # >= 10 clauses: the reference's own sum order is heap-dependent -> 1e-5 relative (north_star)
            parity = bool(np.allclose(g_hits["score"], cs, rtol=1e-5, atol=0) and (g_totals == ct).all())
        else:
            parity = bool((g_hits["doc"] == cd).all() and (g_hits["score"].view(np.int32) == cs.view(np.int32)).all()
                          and (g_totals == ct).all())
        base = {"value": done_q / spent, "unit": "queries/s", "cores": cores, "kind": "port",
                "postings_per_sec": float(sample_postings) * reps / spent,
                "postings_per_sec_per_core": float(sample_postings) * reps / spent / cores,
                "sample": "%d of the batch's %d queries x %d repetitions (%.1f s), one query per thread, %d threads; oracle = C++ "
                          "restatement of Rucene's CPU IndexSearcher (the Rust original cannot be built here)" % (ns, nq, reps, spent, cores)}
        return base, parity

    def roofline(kernel, kernel_ms, algo_bytes, tag):
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        traffic = profiled_traffic(kernel, tag) or {}
        return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic.get("bytes"), "traffic_source": traffic.get("source"), "kernel": kernel, "kernel_ms": kernel_ms,
                "algorithmic_bytes_per_launch": algo_bytes, "frac_vs_measured_copy_6290": achieved / 6290.0}

    def decode_bench(shard, rep, reps):
        """Block-decode microbenchmark (north_star: >= 40 % of HBM peak): every term with df >= 128 of the shard, docs +
        freqs materialised in HBM as i32. SURVEY 8(d): bytes = encoded block + tail bytes in, 8 B per posting out."""
        keep = shard.seg.terms["doc_freq"] >= 128
        sel = np.tile(shard.seg.terms[keep], rep)
        total = int(sel["doc_freq"].sum())
        d_docs = torch.empty((total,), dtype=torch.int32, device="cuda")
        d_freqs = torch.empty((total,), dtype=torch.int32, device="cuda")
        for _ in range(2):
            shard.leaf.segment.decode_terms_device(sel, d_docs.data_ptr(), d_freqs.data_ptr())
        ctx.set_profiling(True)
        ctx.kernel_stats_reset()
        for _ in range(reps):
            shard.leaf.segment.decode_terms_device(sel, d_docs.data_ptr(), d_freqs.data_ptr())
        st = ctx.kernel_stats()["k_decode_terms"]
        ctx.set_profiling(False)
        ctx.kernel_stats_reset()
        ms = st["total_ms"] / st["launches"]
        b = int(shard.enc[keep].sum()) * rep + 8 * total
        del d_docs, d_freqs
        return {"postings": total, "list_repeats": rep, "kernel": "k_decode_terms", "kernel_ms": ms, "postings_per_sec": total / (ms * 1e-3),
                "algorithmic_bytes": b, "working_set_bytes": int(shard.seg.doc_bytes.size) + 8 * total}, b, ms

    # ---- headline: BASELINE configs[1] (or --workload), one 10M-doc shard per rank -----------------------------------------
    k_of = {"term": 10, "and3": 10, "or10": 100}
    k = args.k or k_of[args.workload]
    shard = Shard(args.docs, rank, rank * args.docs, world)
    res = measure(shard, args.workload, k, args.steps, args.warmup, two_streams=True)
    dom_name = DOMINANT[args.workload]
    # the headline is the faster of the two issue modes (both are in `streams`): two alternating streams win on one GPU
    # (the next step's search kernel runs over this step's merge tail), one stream can win when every step ends in a collective
    two_wins = res["ms_two_streams"] <= res["ms_one_stream"]
    ms_per_step = res["ms_two_streams"] if two_wins else res["ms_one_stream"]
    out = {
        "metric": "queries/sec + postings decoded/sec, BM25 10M-doc synthetic",
        "value": world * nq / (ms_per_step * 1e-3),
        "unit": "queries/s (one query evaluated against one %dM-doc segment; x n_gpus shards)" % (args.docs // 1_000_000),
        "queries_per_sec": nq / (ms_per_step * 1e-3),
        "postings_per_sec": world * res["postings"] / (ms_per_step * 1e-3),
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32 decode + f32 BM25",
        "data": "synthetic",
        "config": {
            "workload": WORKLOAD_TEXT[args.workload],
            "docs_per_shard": args.docs, "vocab": args.vocab, "n_queries": nq, "k": k, "doc_format": ".doc v1 (SIMD-BP128)",
            "parallelism": "segment-sharded x%d, RCCL all-gather of per-shard top-k" % world,
            "postings_per_step_per_shard": res["postings"], "index_build_s": round(shard.build_s, 2), "device": ctx.device_name,
            "issue": ("value = two alternating streams (enqueue-only calls; step i+1's search kernel runs over step i's merge tail)" if two_wins
                      else "value = one stream (enqueue-only calls back to back); the two-stream figure is in `streams`"),
        },
        "streams": {"one_stream_ms_per_step": res["ms_one_stream"], "one_stream_queries_per_sec": world * nq / (res["ms_one_stream"] * 1e-3),
                    "two_streams_ms_per_step": res["ms_two_streams"],
                    "one_stream_with_planning_ms_per_step": res["ms_one_stream_with_planning"],
                    "one_stream_with_planning_queries_per_sec": world * nq / (res["ms_one_stream_with_planning"] * 1e-3),
                    "two_streams_with_array_planning_ms_per_step": res["ms_two_streams_with_array_planning"],
                    "two_streams_with_array_planning_queries_per_sec": world * nq / (res["ms_two_streams_with_array_planning"] * 1e-3),
                    "note": ("planning = term resolution, BM25 weights and sim-table handles redone every step on one host thread: "
                             "`with_planning` through GpuIndexSearcher.pack (one Python object per query), `with_array_planning` through "
                             "GpuIndexSearcher.pack_uniform (the batch handed over as an id array; same structs, test_pack.py)")},
        "roofline": roofline(dom_name, res["kernels_ms"].get(dom_name, 0.0), res["algo_bytes"], PROFILE_TAG[args.workload]),
        "kernels_ms_isolated": res["kernels_ms"],
    }
    out["roofline"]["note"] = ("kernel_ms = average launch duration over K steps issued on one stream (HIP events; a separate pass, the timed "
                               "region carries no events). achieved = ALGORITHMIC bytes / kernel_ms: k_search_term skips blocks whose (freq, "
                               "norm rank) frontier bounds them under the top-k threshold (exact: every posting is still counted and the "
                               "top-k equals the oracle's), so the bytes it actually moves are `traffic`, well below the algorithmic bytes")
    if args.force_dist and world == 1:
        print("force-dist: merged == local: %s" % res.get("force_dist_same"), file=sys.stderr)
        if not res.get("force_dist_same"):
            raise SystemExit("force-dist check failed")

    # ---- CPU baseline + parity for the headline, rank 0, N = 1 only ---------------------------------------------------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        base, parity = cpu_baseline_leg(shard, args.workload, k, res, 8.0, nq)
        out["cpu_baseline"] = base
        out["gpu_over_cpu"] = out["queries_per_sec"] / base["value"]
        out["parity_vs_oracle_full_batch"] = parity

    # ---- north_star's other targets, same run (N = 1 only) -------------------------------------------------------------------
    want = set() if args.configs == "none" or world != 1 or dist_mode else \
        ({"and3", "or10", "block_decode", "out_of_cache"} if args.configs == "all" else set(args.configs.split(",")))
    configs = {}
    for kind in ("and3", "or10"):
        if kind not in want or kind == args.workload:
            continue
        kk = k_of[kind]
        steps = max(3, min(args.steps, 10 if kind == "and3" else 4))
        r = measure(shard, kind, kk, steps, 1, two_streams=False, with_planning=False)
        dom = DOMINANT[kind]
        c = {"workload": WORKLOAD_TEXT[kind], "k": kk, "steps": steps, "ms_per_step": r["ms_one_stream"],
             "queries_per_sec": nq / (r["ms_one_stream"] * 1e-3), "postings_per_sec": r["postings"] / (r["ms_one_stream"] * 1e-3),
             "kernels_ms_isolated": r["kernels_ms"]}
        if kind == "and3":
            # scan bytes = every clause's list read fully (what a scan-intersect kernel moves); touched bytes =
            # only the blocks the lead-driven kernel decoded (counted by the kernel itself) + 1 B norm per lead posting
            touched = ctx.and_touched_bytes()
            lead_postings = int(shard.seg.terms["doc_freq"][r["tids"]].min(axis=1).sum())
            c["scan_bytes"] = r["algo_bytes"]
            c["touched_bytes"] = touched + lead_postings + 8 * kk * nq
            c["roofline"] = roofline(dom, r["kernels_ms"].get(dom, 0.0), c["touched_bytes"], PROFILE_TAG[kind])
            c["roofline"]["scan_equivalent_frac"] = r["algo_bytes"] / (r["kernels_ms"].get(dom, 1e9) * 1e-3) / 1e9 / HBM_PEAK_GBS
            c["roofline"]["note"] = "achieved = touched bytes / kernel_ms; scan_equivalent_frac = scan bytes / kernel_ms"
        else:
            c["algorithmic_bytes"] = r["algo_bytes"]
            c["kernels_ms_per_step"] = r["kernels_ms_per_step"]
            c["launches_per_step"] = r["launches_per_step"]
            or_kernels = [n for n in ("k_or_wide", "k_score_terms", "k_or_windows") if n in r["kernels_ms_per_step"]]
            kms = sum(r["kernels_ms_per_step"][n] for n in or_kernels)
            c["roofline"] = roofline(" + ".join(or_kernels), kms, r["algo_bytes"], PROFILE_TAG[kind])
            c["roofline"]["note"] = ("achieved = scan bytes (all ten lists + norms) / summed duration of the OR kernels per batch; ten clauses: "
                                     "k_or_wide (order-free fixed-point accumulation; queries under its precision floor are run again "
                                     "by k_score_terms + k_or_windows)")
        if not args.no_cpu_baseline:
            base, parity = cpu_baseline_leg(shard, kind, kk, r, 4.0, nq if kind == "and3" else 256)
            c["cpu_baseline"] = base
            c["gpu_over_cpu"] = c["queries_per_sec"] / base["value"]
            c["parity_vs_oracle_full_batch"] = parity
        configs[kind] = c
    if "block_decode" in want:
        d, b, ms = decode_bench(shard, 1, 5)
        d["roofline"] = roofline("k_decode_terms", ms, b, PROFILE_TAG["decode"])
        d["note"] = "10M-doc shard, list decoded once per launch: the .doc (67 MB) and the 162 MB of output fit the 256 MiB Infinity Cache only in part; the launch is 62 us long"
        configs["block_decode"] = d
    if "out_of_cache" in want:
        # a shard whose .doc alone exceeds the 256 MiB Infinity Cache: here "fraction of HBM roofline" means HBM
        big = Shard(args.big_docs, 0, 0, 1)
        oc = {"docs": args.big_docs, "doc_file_bytes": int(big.seg.doc_bytes.size), "index_build_s": round(big.build_s, 2)}
        d, b, ms = decode_bench(big, 1, 3)
        d["roofline"] = roofline("k_decode_terms", ms, b, "r02_decode_big")
        oc["block_decode"] = d
        r = measure(big, "term", 10, 5, 1, two_streams=False, with_planning=False)
        oc["term"] = {"workload": WORKLOAD_TEXT["term"], "ms_per_step": r["ms_one_stream"], "queries_per_sec": nq / (r["ms_one_stream"] * 1e-3),
                      "postings_per_sec": r["postings"] / (r["ms_one_stream"] * 1e-3), "kernels_ms_isolated": r["kernels_ms"],
                      "roofline": roofline("k_search_term", r["kernels_ms"].get("k_search_term", 0.0), r["algo_bytes"], "r02_term_big")}
        if not args.no_cpu_baseline:
            base, parity = cpu_baseline_leg(big, "term", 10, r, 3.0, 256)
            oc["term"]["cpu_baseline"] = base
            oc["term"]["parity_vs_oracle_full_batch"] = parity
        configs["out_of_cache"] = oc
    if configs:
        out["configs"] = configs

    sys.stdout.flush()
    os.dup2(stdout_fd, 1)
    os.close(stdout_fd)
    if rank == 0:
        print(json.dumps(out), flush=True)
    os.dup2(2, 1)  # anything native libraries print while shutting down stays off stdout too
    if dist_mode:
        dist.barrier()
        comm.close()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
