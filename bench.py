#!/usr/bin/env python3
"""bench.py — the hot path on BASELINE.json's metric: queries/sec + postings decoded/sec, BM25, 10M-doc Zipfian.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: 1024 single-term BM25 queries (BASELINE.json configs[1],
SURVEY.md §8(d)) PLANNED (term ids -> term states, BM25 weights: the native batch planner behind the C ABI) and evaluated
on the GPU against a 10M-doc Zipfian segment that is already resident in HBM (decode -> BM25 -> top-10). With N GPUs the
index is segment-sharded (one 10M-doc shard per rank, shard = rank), the query batch is replicated, every rank evaluates
it against its shard, per-shard top-k is all-gathered over RCCL and merged on the GPU — weak scaling: the unit is one
query evaluated against one 10M-doc segment. Rank 0 prints ONE JSON line.

What the line claims, and how each claim is kept honest:
  * `value` (queries/s) has the planning inside the timed region;
  * `roofline.frac` is a BANDWIDTH fraction: bytes the dominant kernel really touched (counted by the kernel:
    rgpu_last_search_counters) / its launch duration (HIP events) / 8 TB/s. What an exhaustive scorer would have had to
    stream for the same answers is `scan_equivalent_gbs` — a rate, not a fraction (the TERM kernel skips, exactly, every
    block whose score bound cannot reach the top-k);
  * `postings_decoded_per_sec` counts postings that were unpacked; `postings_covered_per_sec` those the queries span;
  * every timed batch is compared with the CPU oracle on the full batch (N = 1): bit-exact doc ids and scores for TERM /
    AND; for the 10-clause OR — which the reference itself sums in heap order — doc ids are judged by the oracle's own
    score of every returned doc (oracle/parity.py) and `docs_differing` is reported.
Under "configs" (N = 1: all; N > 1: and3, i.e. BASELINE configs[4]'s workload): 3-term AND (configs[2]), 10-term OR top-100
(configs[3]), block decode warm (prepared block store) and COLD (.doc bytes -> skip decode -> postings), the positions side (SURVEY
8(f)3: a materialising .pos decode and two-term exact phrases on the same corpus indexed with positions), and the first four
out of the Infinity Cache on a 100M-doc shard. `--configs none` skips them.
"""
import argparse
import glob
import json
import gc
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
SEED_QUERIES = 0x527563656E65 ^ 0x51  # "Rucene" ^ purpose tag
ROUND = "r06"


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


def term_file_bytes(terms, doc_len_end):
    """Bytes of the .doc file a term owns: postings AND skip data (the next term starts right behind them)."""
    start = terms["doc_start_fp"].astype(np.int64)
    nxt = np.empty_like(start)
    nxt[:-1] = start[1:]
    nxt[-1] = doc_len_end
    out = nxt - start
    out[terms["doc_freq"] <= 1] = 0
    return out.astype(np.int64)


def profiled_traffic(kernel, tag):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC summary profiles/<tag>_rocprofv3_summary.txt:
    2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes. The factor 2 is MI355X_MICROARCH.md's gfx950 correction (FETCH_SIZE
    tallies the 128-byte requests of wide 16-byte-per-lane streaming reads at 64 bytes — these kernels' row loads);
    WRITE_SIZE is taken as reported. None when no such profile is committed — bench.py never runs rocprof."""
    if " + " in kernel:  # several kernels per batch: their per-launch traffic summed
        parts = [profiled_traffic(x, tag) for x in kernel.split(" + ")]
        if any(x is None for x in parts):
            return None
        return {"bytes": sum(x["bytes"] for x in parts), "lower": sum(x["lower"] for x in parts), "source": parts[0]["source"]}
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "%s_rocprofv3_summary.txt" % tag))):
        fetch, write = {}, {}   # per kernel line (a name may stand for several instantiations / kernels: k_skip_ = k_skip_terms, k_skip_dir<1>, <2>, k_skip_groups; k_scan_*): summed
        for line in open(path):
            if kernel not in line:
                continue
            name = line.split("  ")[1] if line.startswith("  ") else line
            m = re.search(r"FETCH_SIZE=([0-9.e+]+)", line)
            if m:
                fetch[name.strip()] = float(m.group(1))
            m = re.search(r"WRITE_SIZE=([0-9.e+]+)", line)
            if m:
                write[name.strip()] = float(m.group(1))
        if fetch and write:
            # `lower`: every request a 64-byte one (FETCH_SIZE as reported) — what profiles/r06_fetch_granule.txt measures for random
            # dword / byte gathers (k_search_and's probes, norm gathers); `bytes`: every request a 128-byte one tallied at 64 (the
            # guide's x 2, right for coalesced row loads). A kernel that mixes both lies in between.
            best = {"bytes": (2.0 * sum(fetch.values()) + sum(write.values())) * 1024.0, "lower": (sum(fetch.values()) + sum(write.values())) * 1024.0,
                    "source": os.path.relpath(path, ROOT)}
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
DOMINANT = {"term": "k_search_term", "and3": "k_search_and", "or10": "k_or_lazy"}
K_OF = {"term": 10, "and3": 10, "or10": 100}


LINE_MAX_BYTES = 8000   # round 5's 25 KB line was not parsed by the driver (round 4's 19 KB one was); the contract keys need < 3 KB

# scalars lifted from the full tree into the ONE stdout line: name -> path in the tree
HOISTED = [
    ("and3_queries_per_sec", "configs.and3.queries_per_sec"), ("and3_ms_per_step", "configs.and3.ms_per_step"),
    ("and3_gpu_over_cpu", "configs.and3.gpu_over_cpu"), ("and3_roofline_frac", "configs.and3.roofline.frac"),
    ("and3_kernel_ms", "configs.and3.roofline.kernel_ms"), ("and3_kernel_ms_suspect", "configs.and3.roofline.kernel_ms_suspect"),
    ("and3_traffic_over_touched", "configs.and3.roofline.traffic_over_bytes"), ("and3_traffic_lower_over_touched", "configs.and3.roofline.traffic_lower_over_bytes"),
    ("and3_parity_vs_oracle", "configs.and3.parity_vs_oracle"),
    ("or10_queries_per_sec", "configs.or10.queries_per_sec"), ("or10_roofline_frac", "configs.or10.roofline.frac"),
    ("or10_deferred_queries_per_sec", "configs.or10.deferred.queries_per_sec"),
    ("or10_parity_vs_oracle", "configs.or10.parity_vs_oracle"), ("or10_docs_differing", "configs.or10.parity.docs_differing"),
    ("block_decode_frac", "configs.block_decode.roofline.frac"), ("cold_frac", "configs.cold.roofline.frac"),
    ("cold_scored_frac", "configs.cold.cold_scored.frac"), ("cold_wall_ms", "configs.cold.wall_ms_incl_host_planning"),
    ("phrase2_queries_per_sec", "configs.positions.phrase2.queries_per_sec"), ("sloppy2_queries_per_sec", "configs.positions.sloppy2.queries_per_sec"),
    ("phrase2_parity_vs_oracle", "configs.positions.phrase2.parity_vs_oracle"), ("sloppy2_parity_vs_oracle", "configs.positions.sloppy2.parity_vs_oracle"),
    ("big_block_decode_frac", "configs.out_of_cache.block_decode.roofline.frac"), ("big_cold_frac", "configs.out_of_cache.cold.roofline.frac"),
    ("big_cold_scored_frac", "configs.out_of_cache.cold.cold_scored.frac"),
    ("big_term_queries_per_sec", "configs.out_of_cache.term.queries_per_sec"), ("big_term_roofline_frac", "configs.out_of_cache.term.roofline.frac"),
    ("big_and3_queries_per_sec", "configs.out_of_cache.and3.queries_per_sec"), ("big_and3_roofline_frac", "configs.out_of_cache.and3.roofline.frac"),
    ("big_and3_kernel_ms", "configs.out_of_cache.and3.roofline.kernel_ms"),
    ("big_and3_traffic_over_touched", "configs.out_of_cache.and3.roofline.traffic_over_bytes"),
    ("big_and3_traffic_lower_over_touched", "configs.out_of_cache.and3.roofline.traffic_lower_over_bytes"),
    ("big_and3_parity_vs_oracle", "configs.out_of_cache.and3.parity_vs_oracle"),
    ("big_or10_queries_per_sec", "configs.out_of_cache.or10.queries_per_sec"), ("big_or10_roofline_frac", "configs.out_of_cache.or10.roofline.frac"),
    ("big_or10_parity_vs_oracle", "configs.out_of_cache.or10.parity_vs_oracle"), ("big_or10_parity_queries", "configs.out_of_cache.or10.parity.queries_checked"),
    ("sharded_ratio_two_streams", "sharded_overhead.ratio_two_streams"), ("sharded_ratio_one_stream", "sharded_overhead.ratio_one_stream"),
    ("sharded_forced_gather_ratio_two_streams", "sharded_overhead.forced_gather.ratio_two_streams"),
    ("sharded_forced_gathers_issued", "sharded_overhead.forced_gather.gathers_issued"),
    ("sharded_forced_gather_same_rows", "sharded_overhead.forced_gather.same_rows"),
    ("one_stream_ms_per_step", "one_stream_ms_per_step"), ("two_calls_ms_per_step", "streams.ms_two_calls_two_streams"),
    ("fused_same_rows_as_two_calls", "fused_same_rows_as_two_calls"), ("resident_plan_ms_per_step", "streams.ms_resident_plan_two_streams"),
    ("kernel_plus_merge_ms", "kernel_plus_merge_ms"), ("step_over_kernels", "step_over_kernels"),
    ("sustained_queries_per_sec", "sustained.queries_per_sec"), ("sustained_ms_per_step", "sustained.ms_per_step"), ("sustained_seconds", "sustained.seconds"),
]
LINE_REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config", "roofline")


def _dig(tree, path):
    node = tree
    for part in path.split("."):
        if not isinstance(node, dict) or part not in node:
            return None
        node = node[part]
    return node


def _short(text, n):
    text = str(text)
    return text if len(text) <= n else text[:n - 3] + "..."


def _num(v):
    """floats at 5 significant digits: the line is read by people and a parser, the full precision lives in the detail file"""
    if isinstance(v, bool) or not isinstance(v, float):
        return v
    return float("%.5g" % v)


def final_line(full, detail_path="bench_detail.json"):
    """The ONE stdout line of the contract, from the full result tree: the contract's keys, `roofline` and `cpu_baseline` as
    small objects, the north-star's other targets as top-level scalars — and nothing else (the tree itself goes to
    `detail_path` and to stderr). Guaranteed: strict JSON that round-trips, < LINE_MAX_BYTES."""
    line = {}
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        line[key] = _num(full.get(key))
    cfg = full.get("config", {})
    line["config"] = {k: (_short(v, 160) if isinstance(v, str) else v) for k, v in cfg.items()
                      if k in ("workload", "docs_per_shard", "vocab", "n_queries", "k", "doc_format", "parallelism", "device", "timed_regions", "issue")}
    roof = full.get("roofline") or {}
    line["roofline"] = {k: (_short(roof[k], 200) if isinstance(roof.get(k), str) else _num(roof.get(k)))
                        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "kernel_ms_suspect", "bytes_per_launch",
                                  "bytes_are", "traffic_source", "across_rounds") if k in roof}
    if "cpu_baseline" in full:
        cb = full["cpu_baseline"]
        line["cpu_baseline"] = {"value": _num(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                                "sample": _short(cb.get("sample", ""), 240)}
    if "parity" in full:
        par = full["parity"]
        line["parity"] = {"ok": full.get("parity_vs_oracle"), "queries_checked": par.get("queries_checked"), "queries_in_batch": par.get("queries_in_batch"),
                          "sampled": par.get("sampled"), "rule": _short(par.get("rule", ""), 120)}
    for key in ("queries_per_sec", "postings_decoded_per_sec", "postings_covered_per_sec", "gpu_over_cpu", "parity_vs_oracle", "ms_per_step_min_median_max",
                "whole_index_docs", "whole_index_queries_per_sec", "without_sketches_queries_per_sec", "force_dist_same", "gathers_issued"):
        if full.get(key) is not None:
            v = full[key]
            line[key] = [_num(x) for x in v] if isinstance(v, list) else _num(v)
    lat = full.get("latency_batch_of_one")
    if lat:
        # [p50, p99] in us of rgpu_search_batch(n_queries = 1), host buffers, blocking; then the same with planning inside
        line["latency_us_p50_p99"] = {kind: [_num(v["p50_us"]), _num(v["p99_us"])] for kind, v in lat.items() if isinstance(v, dict) and "p50_us" in v}
        line["latency_planned_us_p50_p99"] = {kind: [_num(v["planned_p50_us"]), _num(v["planned_p99_us"])] for kind, v in lat.items() if isinstance(v, dict) and "planned_p50_us" in v}
        line["latency_same_as_batch"] = all(v.get("same_as_batch", True) for v in lat.values() if isinstance(v, dict))
    for name, path in HOISTED:
        v = _dig(full, path)
        if v is not None:
            line[name] = _num(v)
    line["detail"] = detail_path
    text = json.dumps(line, allow_nan=False)
    # never over the cap: the hoisted scalars go first (last one first), the contract keys stay
    names = [n for n, _ in HOISTED if n in line]
    while len(text.encode()) >= LINE_MAX_BYTES and names:
        line.pop(names.pop())
        line["truncated"] = True
        text = json.dumps(line, allow_nan=False)
    assert len(text.encode()) < LINE_MAX_BYTES, "the final line does not fit %d bytes" % LINE_MAX_BYTES
    assert "\n" not in text and json.loads(text) == json.loads(json.dumps(line)), "the final line does not round-trip"
    missing = [k for k in LINE_REQUIRED if k not in line or (line[k] is None and k != "vs_baseline")]
    missing += ["roofline." + k for k in ("bound", "achieved", "peak", "unit", "frac", "traffic") if k not in line["roofline"]]
    assert not missing, "the final line lacks %s" % missing
    return text


def _jsonable(o):
    if isinstance(o, dict):
        return {str(k): _jsonable(v) for k, v in o.items() if not str(k).startswith("_")}
    if isinstance(o, (list, tuple)):
        return [_jsonable(v) for v in o]
    if isinstance(o, np.generic):
        return o.item()
    if isinstance(o, np.ndarray):
        return o.tolist()
    if isinstance(o, float) and (o != o or o in (float("inf"), float("-inf"))):
        return None
    return o


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
    ap.add_argument("--configs", default="all", help="all | none | comma list of and3,or10,latency,block_decode,cold,positions,out_of_cache (N > 1: and3 only)")
    ap.add_argument("--big-docs", type=int, default=100_000_000, help="size of the out-of-cache shard")
    ap.add_argument("--regions", type=int, default=5, help="timed regions of --steps steps per figure; the median region is reported")
    ap.add_argument("--detail", default="", help="where the full result tree goes (default: bench_detail.json next to bench.py, and gpurun_out/ when present)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true",
                    help="developer check: take the N > 1 code path (RCCL all-gather + device merge) with a world of one")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N > 1)" % (args.gpus, world))

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for RCCL across processes (before the HSA runtime starts)
    import torch
    import torch.distributed as dist
    import rucene_amd
    from rucene_amd import indexgen, _lib

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
    ctx = rucene_amd.Context(device=local_rank, profile_kernels=False)
    from rucene_amd import dist as rdist
    # N > 1: the collective lives behind the C ABI (rgpu_comm_*: RCCL linked into librucene_gpu.so); torch.distributed only
    # carries the 128-byte communicator id, the barriers around the timed region and the max-over-ranks of the time
    comm = rdist.create_comm(ctx) if dist_mode else None
    OPS = {"term": _lib.OP_TERM, "and3": _lib.OP_AND, "or10": _lib.OP_OR}

    class Shard:
        """One segment resident in HBM + the searcher over it (statistics of shard 0, the first largest leaf)."""

        def __init__(self, docs, shard, doc_base, n_shards):
            t0 = time.time()
            self.seg = indexgen.build_zipf(docs, args.vocab, shard=shard, doc_base=doc_base)
            self.build_s = time.time() - t0
            self.leaf = rucene_amd.LeafReader.from_synthetic(self.seg)
            t0 = time.perf_counter()
            self.searcher = rucene_amd.GpuIndexSearcher([self.leaf], ctx=ctx)
            self.upload_s = time.perf_counter() - t0   # rgpu_segment_upload: header checks + .doc, norms over PCIe
            stats_seg = self.seg if shard == 0 else indexgen.build_zipf(docs, args.vocab, shard=0)
            self.searcher.override_statistics(rucene_amd.CollectionStatistics("body", 0, docs * n_shards, stats_seg.doc_count,
                                                                              stats_seg.sum_total_term_freq),
                                              None if shard == 0 else stats_seg.terms)
            self.enc = term_encoded_bytes(self.seg.terms, self.seg.doc_bytes.size - 16)

        def batch(self, kind, k):
            tids = build_queries(nq, kind, SEED_QUERIES)
            packed = self.searcher.pack_uniform(OPS[kind], tids, self.leaf)   # the native planner (rgpu_plan_uniform_ids)
            flat = tids.reshape(-1)
            postings = int(self.seg.terms["doc_freq"][flat].sum())
            # encoded blocks + tails + 1 B norm per scored posting + 8 k B output per query
            algo_bytes = int(self.enc[flat].sum()) + postings + 8 * k * nq
            return tids, packed, postings, algo_bytes

    class Lane:
        def __init__(self, k):
            self.stream = torch.cuda.Stream()
            self.hits = torch.empty((nq, k), dtype=torch.int64, device="cuda")      # rgpu_hit {i32 doc, f32 score}
            self.totals = torch.empty((nq,), dtype=torch.int64, device="cuda")
            self.local_hits = torch.empty((nq, k), dtype=torch.int64, device="cuda")  # this rank's shard alone (parity checks)
            self.local_totals = torch.empty((nq,), dtype=torch.int64, device="cuda")

    def measure(shard, kind, k, steps, warmup, full):
        """Times `steps` passes of one batch. rgpu_search_batch_device only enqueues (staging copy + kernels): back-to-back
        steps overlap the host-side work of batch i+1 with the kernels of batch i; every timed region is bracketed by a
        barrier + device-wide synchronize on both sides and runs WITHOUT per-kernel events. N > 1: a step =
        rgpu_search_batch_sharded: search -> ONE RCCL all-gather of the per-shard {top-k, count, status} records -> device
        merge, enqueued in order on one stream without host syncs.
        Figures: "planned" = the batch is planned (ids -> rgpu_query_term[]) again in every step, natively; "resident
        plan" = planned once. `full` adds the one-stream and object-planner variants of the headline table."""
        tids, packed, postings, algo_bytes = shard.batch(kind, k)
        lanes = [Lane(k), Lane(k)]
        state = {"n": 0}
        merged = {}
        T, B = rucene_amd.TermQuery, rucene_amd.BooleanQuery

        def step_fused(n_lanes):
            """the serving step: term ids -> rows, ONE call behind the C ABI (rgpu_planner_search_uniform_ids_device / _sharded)"""
            lane = lanes[state["n"] % n_lanes]
            state["n"] += 1
            merged["hits"], merged["totals"] = lane.hits, lane.totals
            shard.searcher.search_uniform_device(OPS[kind], tids, shard.leaf, k, lane.hits.data_ptr(), lane.totals.data_ptr(), lane.stream.cuda_stream,
                                                 comm=comm if dist_mode else None)

        def step(pk, n_lanes):
            lane = lanes[state["n"] % n_lanes]
            state["n"] += 1
            merged["hits"], merged["totals"] = lane.hits, lane.totals
            if dist_mode:
                comm.search_batch_sharded(shard.leaf.segment, pk[0], pk[1], k, lane.hits.data_ptr(), lane.totals.data_ptr(), lane.stream.cuda_stream)
            else:
                shard.leaf.segment.search_batch_device(pk[0], pk[1], k, lane.hits.data_ptr(), lane.totals.data_ptr(), lane.stream.cuda_stream)

        def timed(n_lanes, replan, n_steps=steps, regions=None):
            """R timed regions of exactly n_steps steps each (every one bracketed by barrier + device synchronize on both
            sides, warmup in front of the first): {"min", "median", "max"} ms per step. The figure reported is the MEDIAN
            region — a region of K = 20 steps of 0.06 ms is 1.3 ms long, and one such region is box lottery."""
            per = [timed_region(n_lanes, replan, n_steps, warmup) for r in range(regions or args.regions)]
            per.sort()
            return {"min": per[0], "median": per[len(per) // 2] if len(per) & 1 else 0.5 * (per[len(per) // 2 - 1] + per[len(per) // 2]),
                    "max": per[-1], "regions": len(per), "steps_per_region": n_steps}

        def timed_region(n_lanes, replan, n_steps, n_warm):
            qs = None
            if replan == "objects":
                if kind == "term":
                    qs = [T(int(t[0])) for t in tids]
                elif kind == "and3":
                    qs = [B.build([T(int(x)) for x in t], []) for t in tids]
                else:
                    qs = [B.build([], [T(int(x)) for x in t]) for t in tids]
            # (K = 20 steps of 0.04 ms are a 0.8 ms region: one collection of the interpreter's garbage inside it is a third of it.
            # The collection happens IN FRONT of the warm-up steps: until round 6 it sat between them and the timed region, and the
            # tens of milliseconds it takes are long enough for an idle MI355X to drop its clocks — the same 20 calls measured 48-51 us
            # per step behind a collection and 36-37 us behind the warm-up steps, scripts/host_call_probe.py. The warm-up steps are
            # the step being timed.)
            gc.collect()
            gc.disable()
            for _ in range(n_warm):
                if replan == "fused":
                    step_fused(n_lanes)
                else:
                    step(packed, n_lanes)
            torch.cuda.synchronize()
            if dist_mode:
                dist.barrier()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(n_steps):
                if replan == "fused":
                    step_fused(n_lanes)
                    continue
                if replan == "objects":
                    pk = shard.searcher.pack(qs, shard.leaf)
                elif replan == "array":
                    pk = shard.searcher.pack_uniform(OPS[kind], tids, shard.leaf)
                else:
                    pk = packed
                step(pk, n_lanes)
            torch.cuda.synchronize()
            if dist_mode:
                dist.barrier()
            torch.cuda.synchronize()
            el = time.perf_counter() - t
            gc.enable()
            if dist_mode:
                tt = torch.tensor([el], dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                el = float(tt.item())
            return 1e3 * el / n_steps

        res = {"tids": tids, "postings": postings, "algo_bytes": algo_bytes}
        # untimed, in front of everything: 15-20 ms of the workload itself, so that the first timed region (W warmup steps + K steps
        # can be 2 ms in all) does not start on a GPU that is still raising its clocks — the same batch measured 0.076 ms per step
        # as the first thing a process did and 0.064 a few seconds later
        # (a fixed number of steps, not a time: with N > 1 every step is a collective, and the ranks must make the same calls)
        for _ in range({"term": 256, "and3": 64}.get(kind, 4)):
            step(packed, 2)
        step_fused(2)
        torch.cuda.synchronize()
        fused_same = True
        if not dist_mode:   # the fused call's rows are the two calls' rows
            a, b = lanes[0], lanes[1]
            shard.leaf.segment.search_batch_device(packed[0], packed[1], k, a.hits.data_ptr(), a.totals.data_ptr(), a.stream.cuda_stream)
            shard.searcher.search_uniform_device(OPS[kind], tids, shard.leaf, k, b.hits.data_ptr(), b.totals.data_ptr(), b.stream.cuda_stream)
            ctx.synchronize()
            torch.cuda.synchronize()
            fused_same = bool(torch.equal(a.hits, b.hits)) and bool(torch.equal(a.totals, b.totals))
        res["fused_same_rows_as_two_calls"] = fused_same
        res["regions"] = {"ms_planned_two_streams": timed(2, "fused"), "ms_planned_one_stream": timed(1, "fused")}
        if full:
            res["regions"]["ms_two_calls_two_streams"] = timed(2, "array")
            res["regions"]["ms_two_calls_one_stream"] = timed(1, "array")
            res["regions"]["ms_resident_plan_two_streams"] = timed(2, False)
            res["regions"]["ms_resident_plan_one_stream"] = timed(1, False)
            res["regions"]["ms_object_planner_one_stream"] = timed(1, "objects", max(3, steps // 4), 3)
        for name, reg in res["regions"].items():
            res[name] = reg["median"]
        if full and not dist_mode:
            # ... and the same step for about 2.5 s on end (the regions above are 0.5 ms long each: a GPU-busy sampler never sees them, and
            # clocks / thermals have no time to settle): the sustained rate next to the median region. One process, so the number of
            # steps may follow this box's step time.
            n_sus = int(min(400_000, max(1000, 2500.0 / res["ms_planned_two_streams"])))
            gc.collect()
            gc.disable()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(n_sus):
                step_fused(2)
            torch.cuda.synchronize()
            el = time.perf_counter() - t
            gc.enable()
            res["sustained"] = {"steps": n_sus, "seconds": el, "ms_per_step": 1e3 * el / n_sus, "queries_per_sec": nq * n_sus / el,
                                "note": "the headline's planned two-stream step back to back, one timed region of `steps` steps"}
        # isolated kernel durations: the same steps (the call the timed regions make) on ONE stream with HIP events around every launch
        ctx.set_profiling(True)
        ctx.kernel_stats_reset()
        for _ in range(steps):
            step_fused(1)
        torch.cuda.synchronize()
        stats = ctx.kernel_stats()
        # MEDIAN launch duration (rgpu_kernel_stat.median_ms): round 5's mean over K launches read 1.775 ms on the driver's box for a
        # kernel rocprofv3 times at 0.228 ms — one stalled launch multiplies a mean
        res["kernels_ms"] = {n: s["median_ms"] for n, s in stats.items() if s["timed_launches"] > 0}
        res["kernels_ms_min_max_mean"] = {n: [s["min_ms"], s["max_ms"], s["total_ms"] / max(1, s["timed_launches"])] for n, s in stats.items() if s["timed_launches"] > 0}
        res["launches_per_step"] = {n: s["launches"] / steps for n, s in stats.items() if s["timed_launches"] > 0}
        res["kernels_ms_per_step"] = {n: s["median_ms"] * s["launches"] / steps for n, s in stats.items() if s["timed_launches"] > 0}
        res["counters"] = ctx.last_search_counters()   # of the last launch: what it decoded vs what its queries cover
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

    def physical_cores():
        """Threads for the CPU leg: one per physical core THIS PROCESS MAY USE — the affinity mask and a cgroup CPU quota count (a
        GPU box hands a container 256 logical CPUs and a quota of a few dozen: 128 threads then time-slice, and the per-thread
        rate says nothing about the code)."""
        try:
            import psutil
            n = int(psutil.cpu_count(logical=False) or cores)
        except Exception:
            n = cores
        try:
            n = min(n, len(os.sched_getaffinity(0)))
        except Exception:
            pass
        quota = None
        try:
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
        except Exception:
            try:
                q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    quota = q / per
            except Exception:
                pass
        if quota is not None:
            n = max(1, min(n, int(quota + 0.5)))
        return n, quota

    def cpu_baseline_leg(shard, kind, k, res, budget_s, sample_queries, parity_queries):
        """cpu_baseline: the oracle (C++ restatement of Rucene's CPU IndexSearcher), one query per thread, one thread per PHYSICAL core.
        The timed sample is ONE call over a long queue — the batch's first `sample_queries` queries replicated until every thread has
        ~16 of them to pull (dynamic scheduling: the queue is much longer than the thread count, so the makespan is throughput, not the
        longest query; round 4 timed 1024 queries on 256 threads and measured the Zipf head's tail) — next to a single-thread figure on
        every 16th query. Parity of the GPU's rows on `parity_queries` of the batch (canonical tie rule); fewer than the batch is
        reported as "sampled": true. A reported baseline, not the target."""
        from oracle import binding as orc  # the checker: only ever imported here, after the timed regions
        from oracle import parity
        osearcher = orc.Searcher([orc.Segment(shard.seg.doc_bytes, shard.seg.norms, shard.seg.max_doc, shard.seg.terms,
                                              sum_total_term_freq=shard.seg.sum_total_term_freq)])
        tids = res["tids"]
        op = {"term": orc.OP_TERM, "and3": orc.OP_AND, "or10": orc.OP_OR}[kind]
        threads, quota = physical_cores()
        npq = min(nq, parity_queries)
        ops = np.full(npq, op, np.int32)
        offs = (np.arange(npq + 1) * tids.shape[1]).astype(np.int32)
        cd, cs, cc, ct, _, parity_secs = osearcher.search_batch(ops, offs, np.ascontiguousarray(tids[:npq]).reshape(-1), k, tie_mode=orc.TIE_CANONICAL, threads=cores)
        ns = min(nq, sample_queries)
        base_flat = np.ascontiguousarray(tids[:ns]).reshape(-1)
        sample_postings = int(shard.seg.terms["doc_freq"][base_flat].sum())
        spent, done_q, reps, single = 0.0, 0, 0, None
        sweep = None
        if budget_s > 0:
            # how many threads this box really gives the process: a short sweep (the whole sample once per width), best width wins
            sweep = {}
            ops1 = np.full(ns, op, np.int32)
            offs1 = (np.arange(ns + 1) * tids.shape[1]).astype(np.int32)
            for t in sorted({threads, max(1, threads // 2), max(1, threads // 4), min(threads, 32), min(threads, 16)}):
                _, _, _, _, _, st = osearcher.search_batch(ops1, offs1, base_flat, k, tie_mode=orc.TIE_RUST_HEAP, threads=t)
                sweep[t] = ns / st
            threads = max(sweep, key=lambda t: sweep[t])
            # size the queue from the best width's rate: ~budget_s seconds of work, at least 16 queries per thread
            parity_secs = ns / sweep[threads] * (npq / ns)
            rate = npq / max(parity_secs, 1e-6)
            reps = int(max(1, min(64, max(np.ceil(16.0 * threads / ns), rate * budget_s / ns))))
            flat = np.tile(base_flat, reps)
            ops_r = np.full(ns * reps, op, np.int32)
            offs_r = (np.arange(ns * reps + 1) * tids.shape[1]).astype(np.int32)
            _, _, _, _, _, spent = osearcher.search_batch(ops_r, offs_r, flat, k, tie_mode=orc.TIE_RUST_HEAP, threads=threads)
            done_q = ns * reps
            # one thread, every 16th query of the sample (the same mix of list lengths)
            sub = np.ascontiguousarray(tids[:ns:16])
            _, _, _, _, _, s1 = osearcher.search_batch(np.full(sub.shape[0], op, np.int32), (np.arange(sub.shape[0] + 1) * tids.shape[1]).astype(np.int32),
                                                       sub.reshape(-1), k, tie_mode=orc.TIE_RUST_HEAP, threads=1)
            single = {"queries_per_sec": sub.shape[0] / s1, "postings_per_sec": float(shard.seg.terms["doc_freq"][sub.reshape(-1)].sum()) / s1,
                      "sample": "%d queries (every 16th of the sample), %.2f s" % (sub.shape[0], s1)}
        else:  # budget 0 (the slow 100M-doc legs): the parity pass is the timing sample too — one pass of the oracle, not two
            spent, done_q, reps, ns, threads = parity_secs, npq, 1, npq, cores
            sample_postings = int(shard.seg.terms["doc_freq"][np.ascontiguousarray(tids[:npq]).reshape(-1)].sum())
        g_hits, g_totals = res["g_hits"][:npq], res["g_totals"][:npq]
        parity_info = {"queries_checked": npq, "queries_in_batch": nq, "sampled": npq < nq, "rule": "bit-exact doc ids, score bits and hit counts"}
        if kind == "or10":
            # >= 10 clauses: the reference's own sum order is heap-dependent -> scores within 1e-5 relative (north_star), doc
            # ids judged by the oracle's own score of every returned doc + nothing above the k-th score band missing
            parity_info["rule"] = ("TIE-BAND RULE, not bit-exact: hit counts equal; every returned doc matches and the ORACLE scores it within 1e-5 of the "
                                   "returned score; every oracle hit above the k-th score band is returned (oracle/parity.py); docs_differing = "
                                   "returned docs that are not in the oracle's row")
            try:
                parity_info["docs_differing"] = parity.check_heap_order_batch(osearcher, op, tids[:npq], g_hits, g_totals, cd, cs, cc, ct, rtol=1e-5, what=kind)
                parity_info["docs_returned"] = int(cc.sum())
                ok = bool(np.allclose(g_hits["score"], cs, rtol=1e-5, atol=0))
            except parity.HeapOrderParityError as e:
                parity_info["failure"] = str(e)
                ok = False
        else:
            ok = bool((g_hits["doc"] == cd).all() and (g_hits["score"].view(np.int32) == cs.view(np.int32)).all() and (g_totals == ct).all())
        pps = float(sample_postings) * reps / spent
        base = {"value": done_q / spent, "unit": "queries/s", "cores": threads, "kind": "port",
                "postings_per_sec": pps, "postings_per_sec_per_thread": pps / threads, "logical_cpus": cores, "cgroup_cpu_quota": quota,
                "sample": "%d of the batch's %d queries x %d in ONE queue (%.1f s), one query per thread, %d threads (one per physical core this process may use; %d logical "
                          "CPUs); oracle = C++ restatement of Rucene's CPU IndexSearcher (the Rust original cannot be built here)" % (ns, nq, reps, spent, threads, cores)}
        if sweep is not None:
            base["thread_sweep_queries_per_sec"] = {str(t): v for t, v in sorted(sweep.items())}
        if single is not None:
            base["single_thread"] = single
            ratio = base["postings_per_sec_per_thread"] / single["postings_per_sec"]
            base["per_thread_vs_single_thread"] = ratio
            if ratio < 0.5:
                base["scaling_note"] = ("per-thread throughput at full width is below half the single-thread figure: %d threads share the memory "
                                        "channels / L3 of a multi-socket box whose index lives on one NUMA node, or the process is held to fewer CPUs than "
                                        "the affinity mask and cgroup quota let this script see" % threads)
        return base, ok, parity_info

    def roofline(kernel, kernel_ms, touched_bytes, scan_bytes, tag, what):
        """frac = bytes the kernel really had to touch / its launch duration / peak. `scan_equivalent_gbs`: the rate at which
        an exhaustive scorer would have had to stream the lists for the same answers (NOT a bandwidth claim)."""
        achieved = touched_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        traffic = profiled_traffic(kernel, tag) or {}
        r = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
             "traffic": traffic.get("bytes"), "traffic_source": traffic.get("source"), "kernel": kernel, "kernel_ms": kernel_ms,
             "bytes_per_launch": touched_bytes, "bytes_are": what, "frac_vs_measured_copy_6290": achieved / 6290.0}
        if traffic.get("bytes") and touched_bytes > 0:
            r["traffic_over_bytes"] = traffic["bytes"] / touched_bytes   # well above 1 = wasted re-reads
            r["traffic_lower_bound"] = traffic["lower"]                  # FETCH_SIZE uncorrected (gathers: profiles/r06_fetch_granule.txt)
            r["traffic_lower_over_bytes"] = traffic["lower"] / touched_bytes
        if scan_bytes is not None and scan_bytes != touched_bytes:
            r["scan_bytes_per_launch"] = scan_bytes
            r["scan_equivalent_gbs"] = scan_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        assert r["frac"] <= 1.0, "a bandwidth fraction above 1: %r" % (r,)
        return r

    def or_deferred_leg(shard, k, steps, res):
        """The 10-clause OR batch through a context opened with rgpu_config.or_deferred: rgpu_search_batch_device only enqueues (the
        fixed-point kernels' hand-back flags are looked at by the next call that needs the scratch slot, or by rgpu_synchronize), so
        the host's partitioning of batch i + 1 runs under the kernels of batch i. Planned steps on two alternating streams, ONE
        rgpu_synchronize at the end of the timed region; the rows must be those of the default (blocking-per-group) mode, bit for bit."""
        ctx_d = rucene_amd.Context(device=local_rank, or_deferred=True)
        try:
            leaf = rucene_amd.LeafReader.from_synthetic(shard.seg)
            sd = rucene_amd.GpuIndexSearcher([leaf], ctx=ctx_d)
            tids = res["tids"]
            lanes = [Lane(k), Lane(k)]

            def one(i):
                pk = sd.pack_uniform(OPS["or10"], tids, leaf)
                lane = lanes[i % 2]
                leaf.segment.search_batch_device(pk[0], pk[1], k, lane.hits.data_ptr(), lane.totals.data_ptr(), lane.stream.cuda_stream)
            for i in range(6):   # (a fresh context: each of its four rotating scratch slots allocates on its first use)
                one(i)
            ctx_d.synchronize()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                one(i)
            ctx_d.synchronize()
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / steps
            last = lanes[(steps - 1) % 2]
            g_hits = last.hits.cpu().numpy().view(rucene_amd.HIT_DTYPE).reshape(nq, k)
            same = bool((g_hits["doc"] == res["g_hits"]["doc"]).all() and (g_hits["score"].view(np.int32) == res["g_hits"]["score"].view(np.int32)).all()
                        and (last.totals.cpu().numpy() == res["g_totals"]).all())
            leaf.segment.close()
            return {"ms_per_step": ms, "queries_per_sec": nq / (ms * 1e-3), "same_rows_as_default_mode": same,
                    "issue": "rgpu_config.or_deferred = 1: two alternating streams, every step planned, one rgpu_synchronize after the timed steps"}
        finally:
            ctx_d.close()

    def no_sketch_leg(shard, k, steps, res):
        """The single-term batch through a context opened with RGPU_TERM_SKETCH=0: no block-max sketches (kernels/search_term.hpp) —
        every item of k_search_term starts without a threshold, as in rounds 2-4. Same planned steps on two alternating streams; the
        rows must be those of the default context, bit for bit. Reported next to the headline so that what the sketches buy is on
        the line, not in prose."""
        os.environ["RGPU_TERM_SKETCH"] = "0"
        try:
            ctx_n = rucene_amd.Context(device=local_rank)
        finally:
            del os.environ["RGPU_TERM_SKETCH"]
        try:
            leaf = rucene_amd.LeafReader.from_synthetic(shard.seg)
            sn = rucene_amd.GpuIndexSearcher([leaf], ctx=ctx_n)
            tids = res["tids"]
            lanes = [Lane(k), Lane(k)]

            def one(i):
                lane = lanes[i % 2]
                sn.search_uniform_device(OPS["term"], tids, leaf, k, lane.hits.data_ptr(), lane.totals.data_ptr(), lane.stream.cuda_stream)
            for i in range(6):
                one(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                one(i)
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / steps
            last = lanes[(steps - 1) % 2]
            g_hits = last.hits.cpu().numpy().view(rucene_amd.HIT_DTYPE).reshape(nq, k)
            same = bool((g_hits["doc"] == res["g_hits"]["doc"]).all() and (g_hits["score"].view(np.int32) == res["g_hits"]["score"].view(np.int32)).all()
                        and (last.totals.cpu().numpy() == res["g_totals"]).all())
            # the planned step is bound by the host (planner + enqueue: ~63 us per batch) with or without sketches; what the sketches
            # change shows with the plan resident: the GPU's share of a step
            pk = sn.pack_uniform(OPS["term"], tids, leaf)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                lane = lanes[i % 2]
                leaf.segment.search_batch_device(pk[0], pk[1], k, lane.hits.data_ptr(), lane.totals.data_ptr(), lane.stream.cuda_stream)
            torch.cuda.synchronize()
            ms_resident = 1e3 * (time.perf_counter() - t0) / steps
            ctx_n.set_profiling(True)
            ctx_n.kernel_stats_reset()
            for i in range(steps):
                leaf.segment.search_batch_device(pk[0], pk[1], k, lanes[0].hits.data_ptr(), lanes[0].totals.data_ptr(), lanes[0].stream.cuda_stream)
            torch.cuda.synchronize()
            st = ctx_n.kernel_stats()
            kernel_ms = st["k_search_term"]["median_ms"]
            blocks = ctx_n.last_search_counters()["blocks_decoded"]
            leaf.segment.close()
            return {"ms_per_step": ms, "queries_per_sec": nq / (ms * 1e-3), "same_rows_as_with_sketches": same,
                    "ms_resident_plan_two_streams": ms_resident, "k_search_term_ms": kernel_ms, "blocks_unpacked": blocks,
                    "issue": "RGPU_TERM_SKETCH=0: two alternating streams, every step planned (the fused call)"}
        finally:
            ctx_n.close()

    def search_config(shard, kind, steps, warmup, full, tag, cpu_budget, cpu_sample, parity_queries):
        """One search workload on one shard: throughput with planning in the timed region, the dominant kernel's roofline
        from what it touched, decoded vs covered postings, the CPU leg and parity."""
        k = args.k if (args.k and kind == args.workload) else K_OF[kind]
        r = measure(shard, kind, k, steps, warmup, full)
        two_wins = r["ms_planned_two_streams"] <= r["ms_planned_one_stream"]
        ms = r["ms_planned_two_streams"] if two_wins else r["ms_planned_one_stream"]
        dom = DOMINANT[kind]
        c = r["counters"]
        kms = r["kernels_ms"].get(dom, 0.0)
        reg = r["regions"]["ms_planned_two_streams" if two_wins else "ms_planned_one_stream"]
        out = {"workload": WORKLOAD_TEXT[kind], "k": k, "steps": steps, "ms_per_step": ms,
               "ms_per_step_min_median_max": [reg["min"], reg["median"], reg["max"]], "timed_regions": reg["regions"],
               "queries_per_sec": nq / (ms * 1e-3),
               "postings_covered_per_step": r["postings"], "postings_decoded_per_step": c["postings_decoded"],
               "postings_decoded_per_sec": c["postings_decoded"] / (ms * 1e-3), "postings_covered_per_sec": r["postings"] / (ms * 1e-3),
               "issue": "two alternating streams" if two_wins else "one stream", "fused_same_rows_as_two_calls": r["fused_same_rows_as_two_calls"],
               "kernels_ms_isolated": r["kernels_ms"]}
        if kind == "or10":
            # k_or_lazy walks the clauses without a doc bitmap (k_score_terms decodes them) and reads the others' bitmap words;
            # queries without a bitmap clause, and the ones it hands back, go through k_or_wide (every posting decoded); queries
            # flagged for the f32 floor are run again by the clause-order kernels
            or_kernels = [n for n in ("k_score_terms", "k_or_lazy", "k_or_wide", "k_or_windows") if n in r["kernels_ms_per_step"]]
            kms = sum(r["kernels_ms_per_step"][n] for n in or_kernels)
            out["kernels_ms_per_step"] = r["kernels_ms_per_step"]
            if c["touched_bytes"] > 0:
                out["roofline"] = roofline(" + ".join(or_kernels), kms, c["touched_bytes"] + 8 * k * nq, r["algo_bytes"], tag,
                                           "touched bytes: encoded bytes + 1 B norm per posting of the walked clauses (each distinct term of the batch "
                                           "once), max_doc / 8 B of bitmap words per bitmap clause and query, every posting of the queries "
                                           "k_or_wide took + 8 k B out")
            else:
                out["roofline"] = roofline(" + ".join(or_kernels), kms, r["algo_bytes"], None, tag,
                                           "scan bytes: every clause's encoded blocks + tails + 1 B norm per posting + 8 k B out")
        elif kind == "and3":
            lead_postings = int(shard.seg.terms["doc_freq"][r["tids"]].min(axis=1).sum())
            touched = c["touched_bytes"] + lead_postings + 8 * k * nq
            out["roofline"] = roofline(dom, kms, touched, r["algo_bytes"], tag,
                                       "touched bytes: encoded bytes of every block the kernel decoded (counted by the kernel) + 1 B norm per lead posting + 8 k B out")
        else:
            touched = c["touched_bytes"] + 8 * k * nq
            out["roofline"] = roofline(dom, kms, touched, r["algo_bytes"], tag,
                                       "touched bytes, counted by the kernel: encoded bytes + norms of every block it unpacked + the directory entries it "
                                       "requested (18 B per block of every 64-block chunk it visited, 8 B per chunk frontier, 2 B per sketch entry) + 8 k B out")
            out["roofline"]["pruning"] = "%d of %d FullBlocks unpacked" % (c["blocks_decoded"], int((shard.seg.terms["doc_freq"][r["tids"].reshape(-1)] // 128).sum()))
        # one isolated launch of the dominant kernel cannot take longer than a whole one-stream step that contains it (+ 20 % for the
        # events' own cost): when it does, the event timing is off and the roofline says so instead of being believed
        out["roofline"]["kernel_ms_min_max_mean"] = r["kernels_ms_min_max_mean"].get(dom)
        out["roofline"]["kernel_ms_suspect"] = bool(kind != "or10" and kms > 1.2 * r["ms_planned_one_stream"])
        # the step cannot have moved its bytes faster than the memory system allows
        assert out["roofline"]["bytes_per_launch"] / (ms * 1e-3) / 1e9 <= HBM_PEAK_GBS, "bytes / ms_per_step exceeds the HBM peak"
        if full:
            out["streams"] = {k2: r[k2] for k2 in ("ms_planned_two_streams", "ms_planned_one_stream", "ms_two_calls_two_streams", "ms_two_calls_one_stream",
                                                   "ms_resident_plan_two_streams", "ms_resident_plan_one_stream", "ms_object_planner_one_stream")}
            if "sustained" in r:
                out["sustained"] = r["sustained"]
            out["streams"]["min_median_max"] = {k2: [v["min"], v["median"], v["max"]] for k2, v in r["regions"].items()}
            out["streams"]["note"] = ("ms per step. planned = term ids -> rows in ONE call per step (rgpu_planner_search_uniform_ids_device: term states, BM25 "
                                      "weights, device descriptors, three enqueues) — the headline; two calls = rgpu_plan_uniform_ids then "
                                      "rgpu_search_batch_device (rounds 3-5's headline); resident plan = planned once, searched every step; "
                                      "object planner = one Python query object per query flattened first (GpuIndexSearcher.pack), then the native planner")
        if kind == "or10" and world == 1 and not dist_mode:
            out["deferred"] = or_deferred_leg(shard, k, steps, r)
        if kind == "term" and full and world == 1 and not dist_mode:
            out["without_sketches"] = no_sketch_leg(shard, k, steps, r)
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            base, ok, info = cpu_baseline_leg(shard, kind, k, r, cpu_budget, cpu_sample, parity_queries)
            out["cpu_baseline"] = base
            out["gpu_over_cpu"] = out["queries_per_sec"] / base["value"]
            # (a comparison of fewer queries than the batch holds only counts when it says so: "sampled": true)
            out["parity_vs_oracle"] = bool(ok and (info["queries_checked"] == info["queries_in_batch"] or info.get("sampled") is True))
            out["parity"] = info
        out["_res"] = r
        return out

    def decode_bench(shard, reps, tag):
        """Block-decode microbenchmark, WARM (north_star: >= 40 % of HBM peak): every term with df >= 128 of the shard from
        its prepared block store, docs + freqs materialised in HBM as i32. SURVEY 8(d): bytes = encoded block + tail bytes in,
        8 B per posting out."""
        keep = shard.seg.terms["doc_freq"] >= 128
        sel = shard.seg.terms[keep]
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
        ms = st["median_ms"]
        b = int(shard.enc[keep].sum()) + 8 * total
        del d_docs, d_freqs
        return {"postings": total, "kernel": "k_decode_terms", "kernel_ms": ms, "postings_decoded_per_sec": total / (ms * 1e-3),
                "working_set_bytes": int(shard.seg.doc_bytes.size) + 8 * total, "source": "prepared block store (16-byte aligned rows, decoded tails)",
                "roofline": roofline("k_decode_terms", ms, b, None, tag, "encoded block + tail bytes in + 8 B per posting out")}

    def cold_bench(shard, tag):
        """The reference's own decode path, end to end: a FRESH copy of the segment (nothing prepared) -> skip-list decode +
        block framing + alignment + tail decode + validation (kernels/prepare.hpp stage A: what ForUtil::read_block's header
        parse, read_vint_block and Lucene50SkipReader do per term, for_util.rs:187-243, posting_reader.rs:308-333,
        skip_reader.rs:315-584) -> k_decode_terms, for every term with df >= 128. Bytes = the terms' .doc bytes (postings AND
        skip data) in + 8 B per posting out; time = the kernels' durations summed (HIP events)."""
        keep = shard.seg.terms["doc_freq"] >= 128
        sel = shard.seg.terms[keep]
        total = int(sel["doc_freq"].sum())
        t0 = time.perf_counter()
        seg2 = _lib.Segment(ctx, shard.seg.doc_bytes, shard.seg.norms, shard.seg.max_doc, 0, None)
        upload_s = time.perf_counter() - t0
        d_docs = torch.empty((total,), dtype=torch.int32, device="cuda")
        d_freqs = torch.empty((total,), dtype=torch.int32, device="cuda")
        ctx.set_profiling(True)
        ctx.kernel_stats_reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        seg2.decode_terms_device(sel, d_docs.data_ptr(), d_freqs.data_ptr())   # first touch: stage A of the preparation, then the decode
        torch.cuda.synchronize()
        wall_ms = 1e3 * (time.perf_counter() - t0)
        st = ctx.kernel_stats()
        stage_a = ("k_skip_dir", "k_block_headers", "k_scan_rows", "k_prepare_blocks", "k_decode_terms")
        kms = {n: st[n]["total_ms"] for n in stage_a if n in st}
        ms = sum(kms.values())
        # the same first touch on a store that was released (rgpu_segment_release_prepared_terms): the host's planning and the
        # kernels without the fresh segment's device allocations (hipMalloc of the block store: 0.3 to 20 ms, box to box)
        ctx.set_profiling(False)
        seg2.release_prepared_terms()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        seg2.decode_terms_device(sel, d_docs.data_ptr(), d_freqs.data_ptr())
        torch.cuda.synchronize()
        wall_released_ms = 1e3 * (time.perf_counter() - t0)
        ctx.set_profiling(True)
        # what scoring needs on top (stage B: posting-order norms + block-max frontier words: one norm gather per posting)
        ctx.kernel_stats_reset()
        t0 = time.perf_counter()
        seg2.prepare_terms(sel)
        torch.cuda.synchronize()
        norms_wall_ms = 1e3 * (time.perf_counter() - t0)
        st = ctx.kernel_stats()
        norms_ms = st["k_prepare_norms"]["total_ms"] if "k_prepare_norms" in st else 0.0
        ctx.set_profiling(False)
        ctx.kernel_stats_reset()
        file_bytes = int(term_file_bytes(shard.seg.terms, shard.seg.doc_bytes.size - 16)[keep].sum())
        b = file_bytes + 8 * total
        fp = seg2.footprint()
        held = fp["directory_bytes"] + fp["block_store_bytes"] + fp["posting_norms_bytes"]
        out = {"postings": total, "terms": int(keep.sum()), "kernels_ms": kms, "kernels_ms_total": ms, "wall_ms_incl_host_planning": wall_ms,
               "wall_ms_released_store": wall_released_ms,
               "search_ready_extra": {"k_prepare_norms_ms": norms_ms, "wall_ms": norms_wall_ms,
                                      "note": "posting-order norms + block-max frontier words for scoring (one norm-byte gather per posting); not part of a decode"},
               "postings_decoded_per_sec": total / (ms * 1e-3),
               "doc_file_bytes_of_these_terms": file_bytes, "upload_s_pcie": upload_s,
               "hbm_footprint": fp, "hbm_bytes_held_per_doc_file_byte": (fp["doc_file_bytes"] + held) / max(1, fp["doc_file_bytes"]),
               "roofline": roofline("k_skip_ + k_block_headers + k_scan_ + k_prepare_blocks", ms, b, None, tag,
                                    "the terms' .doc bytes (postings and skip data) in + 8 B per posting out; kernel_ms = the kernels summed")}
        # the first SCORED use of these terms pays stage A (without the materialising decode) AND stage B: reported as one figure
        # (the cold row must not leave k_prepare_norms out). Bytes: the terms' .doc bytes in, their block store
        # out (the same encoded bytes, aligned), one norm byte gathered + one written per posting, 8 B of frontier word per block.
        # A sparse term's norm gather moves a 64-byte sector per posting (profiles/r06_fetch_granule.txt: random sectors arrive at
        # 2.9 TB/s): `sector_bound_ms` = postings x 64 B / 2.9 TB/s is the floor of k_prepare_norms on such lists.
        scored_ms = sum(v for n, v in kms.items() if n != "k_decode_terms") + norms_ms
        enc_bytes = int(shard.enc[keep].sum())
        scored_bytes = file_bytes + enc_bytes + 2 * total + 8 * (total // 128)
        out["cold_scored"] = {"kernels_ms": scored_ms, "k_prepare_norms_ms": norms_ms, "bytes": scored_bytes,
                              "achieved_gbs": scored_bytes / (scored_ms * 1e-3) / 1e9 if scored_ms > 0 else 0.0,
                              "frac": scored_bytes / (scored_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if scored_ms > 0 else 0.0,
                              "norm_gather_sector_bound_ms": total * 64.0 / 2.9e12 * 1e3,
                              "bytes_are": ".doc bytes in + block store out + 1 B norm gathered and 1 B written per posting + 8 B frontier word per block; "
                                           "kernels = k_skip_* + k_block_headers + k_scan_* + k_prepare_blocks + k_prepare_norms"}
        seg2.close()
        del d_docs, d_freqs
        return out

    def positions_bench(docs, n_phrases, k, tag):
        """SURVEY 8(f)3 measured: the same Zipfian postings indexed WITH positions (a posting's `freq` positions start at 0..63 and step
        by 1..16) — (1) `positions_decode`: every position of every term with df >= 128 materialised (rgpu_decode_positions_device:
        BlockPostingIterator::next_position to exhaustion), bytes = those terms' .pos bytes + their freq blocks' share of .doc in,
        4 B per position out; (2) `phrase2`: `n_phrases` two-term exact PhraseQuerys, ranks log-uniform 1..1000, top-k, through
        rgpu_search_phrase_batch (host outputs, blocking: wall time per batch), checked against the oracle's ExactPhraseScorer on
        a sample and timed against it on one core."""
        t0 = time.time()
        seg = indexgen.build_zipf(docs, args.vocab, positions=True)
        build_s = time.time() - t0
        leaf = rucene_amd.LeafReader.from_synthetic_positions(seg)
        searcher = rucene_amd.GpuIndexSearcher([leaf], ctx=ctx)
        out = {"docs": docs, "doc_file_bytes": int(seg.doc_bytes.size), "pos_file_bytes": int(seg.pos_bytes.size), "index_build_s": round(build_s, 2)}
        # ---- (1) positions decode
        keep = seg.terms["doc_freq"] >= 128
        sel, selp = seg.terms[keep], leaf.term_positions[keep]
        total = int(sel["total_term_freq"].sum())
        leaf.segment.attach_positions(leaf.pos_bytes)
        leaf._pos_attached = True
        d_pos = torch.empty((total,), dtype=torch.int32, device="cuda")
        for _ in range(2):
            leaf.segment.decode_positions_device(sel, selp, d_pos.data_ptr())
        ctx.set_profiling(True)
        ctx.kernel_stats_reset()
        reps = 5
        for _ in range(reps):
            leaf.segment.decode_positions_device(sel, selp, d_pos.data_ptr())
        st = ctx.kernel_stats()
        ctx.set_profiling(False)
        ctx.kernel_stats_reset()
        kms = {n: st[n]["total_ms"] / reps for n in ("k_pos_counts", "k_scan_rows", "k_decode_positions") if n in st}
        ms = sum(kms.values())
        nxt = np.empty(seg.terms.size, dtype=np.int64)
        nxt[:-1] = seg.pos_start_fp[1:]
        nxt[-1] = seg.pos_bytes.size - 16
        pos_bytes_in = int((nxt - seg.pos_start_fp)[keep].sum())
        doc_bytes_in = int(term_encoded_bytes(seg.terms, seg.doc_bytes.size - 16)[keep].sum())
        b = pos_bytes_in + 2 * doc_bytes_in + 4 * total   # the freq blocks are unpacked by the count pass and again by the decode pass
        out["positions_decode"] = {"positions": total, "kernels_ms": kms, "kernels_ms_total": ms, "positions_decoded_per_sec": total / (ms * 1e-3),
                                   "roofline": roofline("k_pos_counts + k_scan_ + k_decode_positions", ms, b, None, tag + "_posdec",
                                                        "the terms' .pos bytes + their .doc blocks twice (count pass, decode pass) in + 4 B per position out; kernel_ms = the kernels summed")}
        got = d_pos.cpu().numpy()
        del d_pos
        # ---- (2) two-term phrases: exact (ExactPhraseScorer) and slop 2 (SloppyPhraseScorer) over the same pairs
        ranks = indexgen.log_uniform_ranks(2 * n_phrases, 1, 1000, SEED_QUERIES ^ 0xF2).reshape(-1, 2) - 1
        ix = None
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            from oracle import binding as orc   # the checker, after the timed regions
            ix = orc.PositionsIndex.from_files(seg.doc_bytes, seg.pos_bytes, seg.terms, leaf.term_positions)

        def phrase_leg(slop, what):
            queries = [rucene_amd.PhraseQuery([int(a), int(b2)], slop=slop) for a, b2 in ranks]
            qs, ts = searcher.pack_phrases(queries, leaf)
            for _ in range(2):
                hits, totals = leaf.segment.search_phrase_batch(qs, ts, k)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            steps = 5 if slop == 0 else 3
            for _ in range(steps):
                hits, totals = leaf.segment.search_phrase_batch(qs, ts, k)
            ms_step = 1e3 * (time.perf_counter() - t0) / steps
            ctx.set_profiling(True)
            ctx.kernel_stats_reset()
            leaf.segment.search_phrase_batch(qs, ts, k)
            st = ctx.kernel_stats()
            ctx.set_profiling(False)
            ctx.kernel_stats_reset()
            ph = {"workload": "%d x two-term %s top-%d, ranks log-uniform 1..1000, %d M docs with positions" % (n_phrases, what, k, docs // 1_000_000),
                  "k": k, "slop": slop, "ms_per_step": ms_step, "queries_per_sec": n_phrases / (ms_step * 1e-3), "issue": "blocking call, host outputs",
                  "kernels_ms": {n: v["total_ms"] / max(1, v["launches"]) for n, v in st.items() if v["total_ms"] > 0},
                  "lead_postings_per_step": int(sum(min(int(seg.terms["doc_freq"][a]), int(seg.terms["doc_freq"][b2])) for a, b2 in ranks)),
                  "phrase_hits_per_step": int(totals.sum())}
            if ix is not None:
                # every query of the batch (round 4's launch-size bug sat behind query 358 of a batch whose first 96 were compared)
                n_chk, ok, spent = n_phrases, True, 0.0
                for i in range(n_chk):
                    t0 = time.perf_counter()
                    d, sc, tot = ix.phrase_search([int(ranks[i, 0]), int(ranks[i, 1])], k, seg.norms, seg.max_doc, seg.doc_count, seg.sum_total_term_freq, slop=slop)
                    spent += time.perf_counter() - t0
                    ok = ok and totals[i] == tot and bool((hits[i]["doc"][:d.size] == d).all()) and bool((hits[i]["score"][:d.size].view(np.int32) == sc.view(np.int32)).all())
                ph["parity_vs_oracle"] = bool(ok)
                ph["parity"] = {"queries_checked": n_chk, "queries_in_batch": n_phrases, "sampled": False,
                                "rule": "bit-exact doc ids, score bits and hit counts (%s)" % ("ExactPhraseScorer" if slop == 0 else "SloppyPhraseScorer, two-phase rule with the default next_limit")}
                ph["cpu_baseline"] = {"value": n_chk / spent, "unit": "queries/s", "cores": 1, "kind": "port",
                                      "sample": "the batch's %d queries, one after the other on one core (%.2f s); oracle = C++ restatement of PhraseWeight + %s" % (n_chk, spent, "ExactPhraseScorer" if slop == 0 else "SloppyPhraseScorer")}
                ph["gpu_over_cpu"] = ph["queries_per_sec"] / ph["cpu_baseline"]["value"]
            return ph
        ph = phrase_leg(0, "exact PhraseQuery")
        out["sloppy2"] = phrase_leg(2, "PhraseQuery with slop 2 (SloppyPhraseScorer)")
        if ix is not None:
            # the decoded positions against the oracle's iterator for a handful of terms
            okp = True
            idx = np.nonzero(keep)[0]
            offs = np.concatenate([[0], np.cumsum(sel["total_term_freq"])])
            for j in (4, 40, idx.size // 2, idx.size - 1):   # (lists of <= 500 k docs: the oracle's iterator is driven from Python)
                want = np.array([p for _, _, ps in ix.iterate(int(idx[j]), cap_visits=1 << 20, cap_positions=1 << 23) for p in ps], dtype=np.int32)
                okp = okp and bool((got[offs[j]:offs[j + 1]] == want).all())
            out["positions_decode"]["parity_vs_oracle"] = bool(okp)
            ix.close()
        out["phrase2"] = ph
        leaf.segment.close()
        return out

    def latency_leg(shard, kinds, n_sample):
        """IndexSearcher::search(query, collector) is ONE query per call (search/searcher.rs:487-525): what rust/gpu/searcher.rs
        try_gpu maps to rgpu_search_batch(n_queries = 1) with host buffers, blocking. Per workload: the first `n_sample` queries
        of the bench batch, each as its own call, one after the other on an otherwise idle GPU — p50 / p99 / mean latency of the
        call alone (plan resident) and of plan + call (rgpu_plan_uniform_ids for one query, then the search), and the q/s a
        single caller thread gets that way. The rows of every call are compared with the same query's row of the 1024-query
        batch (bit for bit; >= 10-clause OR: hit counts equal and scores within 1e-5 — the batch and the single call may take
        different fixed-point kernels)."""
        L = _lib.lib()
        out = {}
        for kind in kinds:
            k = K_OF[kind]
            tids = build_queries(nq, kind, SEED_QUERIES)[:n_sample]
            plans = [shard.searcher.pack_uniform(OPS[kind], tids[i:i + 1], shard.leaf) for i in range(tids.shape[0])]
            plans = [(np.ascontiguousarray(q, dtype=_lib.QUERY_DTYPE), np.ascontiguousarray(t, dtype=_lib.QUERY_TERM_DTYPE)) for q, t in plans]
            hits = np.zeros((1, k), dtype=_lib.HIT_DTYPE)
            totals = np.zeros(1, dtype=np.int64)
            rows, tot = np.zeros((tids.shape[0], k), dtype=_lib.HIT_DTYPE), np.zeros(tids.shape[0], dtype=np.int64)
            seg_h = shard.leaf.segment._h

            def call(q, t):
                rc = L.rgpu_search_batch(seg_h, q.ctypes.data, 1, t.ctypes.data, t.size, k, hits.ctypes.data, totals.ctypes.data)
                if rc != 0:
                    raise SystemExit("rgpu_search_batch(n_queries = 1) failed: %d" % rc)
            for q, t in plans:   # untimed: every term of the sample prepared, every kernel variant loaded
                call(q, t)
            gc.collect()
            gc.disable()
            lat = np.empty(len(plans))
            for i, (q, t) in enumerate(plans):
                t0 = time.perf_counter()
                call(q, t)
                lat[i] = time.perf_counter() - t0
                rows[i], tot[i] = hits[0], totals[0]
            lat_p = np.empty(len(plans))
            for i in range(len(plans)):
                t0 = time.perf_counter()
                q, t = shard.searcher.pack_uniform(OPS[kind], tids[i:i + 1], shard.leaf)
                call(np.ascontiguousarray(q, dtype=_lib.QUERY_DTYPE), np.ascontiguousarray(t, dtype=_lib.QUERY_TERM_DTYPE))
                lat_p[i] = time.perf_counter() - t0
            gc.enable()
            us = lambda a, pct: float(np.percentile(a, pct) * 1e6)
            out[kind] = {"calls": int(lat.size), "k": k, "p50_us": us(lat, 50), "p99_us": us(lat, 99), "mean_us": float(lat.mean() * 1e6),
                         "queries_per_sec_one_caller": float(1.0 / lat.mean()),
                         "planned_p50_us": us(lat_p, 50), "planned_p99_us": us(lat_p, 99), "planned_mean_us": float(lat_p.mean() * 1e6),
                         "_rows": rows, "_totals": tot}
        return out

    def latency_parity(lat, kind, batch_hits, batch_totals):
        """a single-query call must answer what the same query answered inside the 1024-query batch"""
        rows, tot = lat.pop("_rows"), lat.pop("_totals")
        n = rows.shape[0]
        same_tot = bool((tot == batch_totals[:n]).all())
        if kind == "or10":
            ok = same_tot and bool(np.allclose(rows["score"], batch_hits["score"][:n], rtol=1e-5, atol=0))
            lat["same_as_batch_rule"] = "hit counts equal, scores within 1e-5 (fixed-point kernels)"
        else:
            ok = same_tot and bool((rows["doc"] == batch_hits["doc"][:n]).all()) and bool((rows["score"].view(np.int32) == batch_hits["score"][:n].view(np.int32)).all())
            lat["same_as_batch_rule"] = "bit-exact doc ids, score bits, hit counts"
        lat["same_as_batch"] = ok
        return lat

    def strip(c):
        c.pop("_res", None)
        return c

    # ---- headline: BASELINE configs[1] (or --workload), one 10M-doc shard per rank -----------------------------------------
    shard = Shard(args.docs, rank, rank * args.docs, world)
    head = search_config(shard, args.workload, args.steps, args.warmup, True, "%s_%s" % (ROUND, args.workload), 8.0, nq, nq)
    res = head["_res"]
    ms_per_step = head["ms_per_step"]
    out = {
        "metric": "queries/sec + postings decoded/sec, BM25 10M-doc synthetic",
        "value": world * nq / (ms_per_step * 1e-3),
        "unit": "queries/s (one query planned and evaluated against one %dM-doc segment; x n_gpus shards)" % (args.docs // 1_000_000),
        "queries_per_sec": nq / (ms_per_step * 1e-3),
        "postings_decoded_per_sec": world * head["postings_decoded_per_sec"],
        "postings_covered_per_sec": world * head["postings_covered_per_sec"],
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32 decode + f32 BM25",
        "data": "synthetic",
        "config": {
            "workload": WORKLOAD_TEXT[args.workload],
            "docs_per_shard": args.docs, "vocab": args.vocab, "n_queries": nq, "k": head["k"], "doc_format": ".doc v1 (SIMD-BP128)",
            "parallelism": "segment-sharded x%d, RCCL all-gather of per-shard top-k" % world,
            "postings_covered_per_step_per_shard": res["postings"], "postings_decoded_per_step_per_shard": head["postings_decoded_per_step"],
            "index_build_s": round(shard.build_s, 2), "segment_upload_s": round(shard.upload_s, 3), "device": ctx.device_name,
            "timed_regions": "%d regions of %d steps, median region reported (min / median / max in ms_per_step_min_median_max)" % (head["timed_regions"], args.steps),
            "issue": "value = %s, every step plans its batch from term ids and enqueues it (one C-ABI call: rgpu_planner_search_uniform_ids_device); inputs (index) resident in HBM" % head["issue"],
        },
        "ms_per_step_min_median_max": head["ms_per_step_min_median_max"],
        "streams": head["streams"],
        "roofline": head["roofline"],
        "kernels_ms_isolated": head["kernels_ms_isolated"],
    }
    if "without_sketches" in head:
        out["without_sketches"] = head["without_sketches"]
        out["without_sketches_queries_per_sec"] = head["without_sketches"]["queries_per_sec"]
        out["without_sketches_same_rows"] = head["without_sketches"]["same_rows_as_with_sketches"]
    out["roofline"]["note"] = ("kernel_ms = average launch duration over K steps issued on one stream (HIP events; a separate pass, the timed "
                               "region carries no events). frac = bytes_per_launch (see bytes_are) / kernel_ms / peak. The north-star's >= 40 % "
                               "block-decode target is configs.block_decode / configs.cold / configs.out_of_cache")
    for key in ("cpu_baseline", "gpu_over_cpu", "parity_vs_oracle", "parity"):
        if key in head:
            out[key] = head[key]
    # the fraction at the headline's OPERATING POINT (two launches co-resident on alternating streams): the same bytes over the
    # step time — next to roofline.frac, which is one isolated launch (kernel_ms may exceed ms_per_step for that reason)
    # a pruned top-k search, not a stream: better pruning lowers the touched bytes and with them `frac` while the launch gets shorter
    out["roofline"]["across_rounds"] = ("pruned search: better pruning lowers touched bytes and frac while the launch gets shorter (r05: 36.0 MB / 0.037 ms = 0.12); compare kernel_ms across rounds; streaming kernel: block_decode_frac")
    out["roofline"]["frac_at_ms_per_step"] = out["roofline"]["bytes_per_launch"] / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS
    # how far the step is from its kernels: every kernel of one step — the plan's copy kernel (round 6), the dominant kernel, the item
    # merge where it is still a launch of its own — as isolated median launches, vs the planned step (two streams overlap them: < 1)
    kp = sum(v for n, v in head["kernels_ms_isolated"].items() if n in (DOMINANT[args.workload], "k_merge_items", "k_merge_lists", "k_stage_copy", "k_stage_term_plan"))
    out["kernel_plus_merge_ms"] = kp
    out["step_over_kernels"] = ms_per_step / kp if kp > 0 else None
    out["step_over_kernels_one_stream"] = head["streams"]["ms_planned_one_stream"] / kp if kp > 0 else None
    out["fused_same_rows_as_two_calls"] = head["fused_same_rows_as_two_calls"]
    out["one_stream_ms_per_step"] = head["streams"]["ms_planned_one_stream"]
    if "sustained" in head:
        out["sustained"] = head["sustained"]
    out["one_stream_queries_per_sec"] = nq / (head["streams"]["ms_planned_one_stream"] * 1e-3)
    # N > 1: the batch is replicated, so the ranks together answer nq queries over an index of world x docs — that rate next to
    # `value` (whose unit is one query against one segment)
    out["whole_index_docs"] = world * args.docs
    out["whole_index_queries_per_sec"] = nq / (ms_per_step * 1e-3)
    if "parity_vs_oracle" in out:
        out["parity_vs_oracle_full_batch"] = out["parity_vs_oracle"]
    if args.force_dist and world == 1:
        print("force-dist: merged == local: %s" % res.get("force_dist_same"), file=sys.stderr)
        if not res.get("force_dist_same"):
            raise SystemExit("force-dist check failed")

    def sharded_overhead(shard, kind, k, steps):
        """What rgpu_search_batch_sharded adds to a step in a world of ONE rank (search into the rank's own region of the gather
        buffer, nothing to gather, the record merge) — the path's own overhead, measured on every N = 1 run (VERDICT r4 item 1:
        0.157 vs 0.093 ms before the in-place record). Planned steps on two alternating streams and on one, local vs sharded."""
        c1 = _lib.Comm(ctx, 1, 0, _lib.comm_unique_id())
        # ... and with the collective REALLY issued (rgpu_config.comm_force_gather: the in-place ncclAllGather of a one-rank
        # communicator + the cross-stream ordering around it) — a second context, since the knob is the context's
        ctx_f = rucene_amd.Context(device=local_rank, comm_force_gather=True)
        leaf_f = rucene_amd.LeafReader.from_synthetic(shard.seg)
        sf = rucene_amd.GpuIndexSearcher([leaf_f], ctx=ctx_f)
        sf.override_statistics(shard.searcher.collection_statistics, None)
        cf = _lib.Comm(ctx_f, 1, 0, _lib.comm_unique_id())
        cf.reserve(nq, k)
        tids = build_queries(nq, kind, SEED_QUERIES)
        lanes = [Lane(k), Lane(k)]
        res = {}
        forced = {}
        for n_lanes in (2, 1):
            def onef(i):
                lane = lanes[i % n_lanes]
                sf.search_uniform_device(OPS[kind], tids, leaf_f, k, lane.hits.data_ptr(), lane.totals.data_ptr(), lane.stream.cuda_stream, comm=cf)
            for i in range(6):
                onef(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                onef(i)
            torch.cuda.synchronize()
            forced["sharded_ms_%s" % ("two_streams" if n_lanes == 2 else "one_stream")] = 1e3 * (time.perf_counter() - t0) / steps
        forced["gathers_issued"] = cf.gathers_issued()
        forced_rows = (lanes[(steps - 1) % 1].hits.clone(), lanes[(steps - 1) % 1].totals.clone())
        for name in ("local", "sharded"):
            for n_lanes in (2, 1):
                def one(i):
                    lane = lanes[i % n_lanes]
                    shard.searcher.search_uniform_device(OPS[kind], tids, shard.leaf, k, lane.hits.data_ptr(), lane.totals.data_ptr(), lane.stream.cuda_stream,
                                                         comm=None if name == "local" else c1)
                for i in range(6):
                    one(i)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(steps):
                    one(i)
                torch.cuda.synchronize()
                res["%s_ms_%s" % (name, "two_streams" if n_lanes == 2 else "one_stream")] = 1e3 * (time.perf_counter() - t0) / steps
        # the merged rows of a world of one are the local rows
        pk = shard.searcher.pack_uniform(OPS[kind], tids, shard.leaf)
        shard.leaf.segment.search_batch_device(pk[0], pk[1], k, lanes[0].hits.data_ptr(), lanes[0].totals.data_ptr(), lanes[0].stream.cuda_stream)
        c1.search_batch_sharded(shard.leaf.segment, pk[0], pk[1], k, lanes[1].hits.data_ptr(), lanes[1].totals.data_ptr(), lanes[1].stream.cuda_stream)
        torch.cuda.synchronize()
        res["same_rows"] = bool(torch.equal(lanes[0].hits, lanes[1].hits)) and bool(torch.equal(lanes[0].totals, lanes[1].totals))
        res["ratio_two_streams"] = res["sharded_ms_two_streams"] / res["local_ms_two_streams"]
        res["ratio_one_stream"] = res["sharded_ms_one_stream"] / res["local_ms_one_stream"]
        res["workload"] = WORKLOAD_TEXT[kind]
        forced["same_rows"] = bool(torch.equal(forced_rows[0], lanes[0].hits)) and bool(torch.equal(forced_rows[1], lanes[0].totals))
        forced["ratio_two_streams"] = forced["sharded_ms_two_streams"] / res["local_ms_two_streams"]
        forced["ratio_one_stream"] = forced["sharded_ms_one_stream"] / res["local_ms_one_stream"]
        forced["note"] = "rgpu_config.comm_force_gather = 1: ncclAllGather (in place, one rank) issued in every step; must have run steps + warmup times"
        assert forced["gathers_issued"] >= 2 * (steps + 6), "the forced all-gather did not run"
        res["forced_gather"] = forced
        c1.close()
        cf.close()
        leaf_f.segment.close()
        ctx_f.close()
        return res

    if world == 1 and not dist_mode:
        out["sharded_overhead"] = sharded_overhead(shard, args.workload, head["k"], 200)

    # ---- north_star's other targets, same run ---------------------------------------------------------------------------------
    if args.configs == "none":
        want = set()
    elif world != 1 or dist_mode:
        want = {"and3"} if args.configs == "all" else (set(args.configs.split(",")) & {"and3", "or10"})   # BASELINE configs[4]'s workload on every N
    else:
        want = {"and3", "or10", "latency", "block_decode", "cold", "positions", "out_of_cache"} if args.configs == "all" else set(args.configs.split(","))
    configs = {}
    batch_rows_extra = {}
    for kind in ("and3", "or10"):
        if kind not in want or kind == args.workload:
            continue
        steps = max(3, min(args.steps, 10 if kind == "and3" else 4))
        c = search_config(shard, kind, steps, 1, False, "%s_%s" % (ROUND, kind), 4.0, nq if kind == "and3" else 256, nq)
        batch_rows_extra[kind] = (c["_res"]["g_hits"], c["_res"]["g_totals"])
        c = strip(c)
        if world > 1:
            c["value_all_shards_queries_per_sec"] = world * c["queries_per_sec"]
            c["whole_index_queries_per_sec"] = c["queries_per_sec"]   # the replicated batch answered over world x docs
        configs[kind] = c
    if "latency" in want:
        # batch of ONE (what IndexSearcher::search maps to): after the batch legs, so that the legs' rows are there to compare with
        lat = latency_leg(shard, ("term", "and3", "or10"), 256)
        batch_rows = {args.workload: (res["g_hits"], res["g_totals"])}
        for kind in ("and3", "or10"):
            if kind in batch_rows_extra:
                batch_rows[kind] = batch_rows_extra[kind]
        for kind in list(lat):
            if kind in batch_rows:
                latency_parity(lat[kind], kind, *batch_rows[kind])
            else:
                lat[kind].pop("_rows"); lat[kind].pop("_totals")
        lat["note"] = ("rgpu_search_batch(n_queries = 1), host buffers, blocking, one call after the other from one thread: p50 / p99 / mean in us "
                       "(plan resident) and planned_* (rgpu_plan_uniform_ids for the one query + the call); 256 calls = the first 256 queries of each batch")
        out["latency_batch_of_one"] = lat
    if "block_decode" in want:
        d = decode_bench(shard, 5, "%s_decode" % ROUND)
        d["note"] = "10M-doc shard: the .doc (67 MB) and the 162 MB of output fit the 256 MiB Infinity Cache only in part; the launch is ~50 us long"
        configs["block_decode"] = d
    if "cold" in want:
        configs["cold"] = cold_bench(shard, "%s_cold" % ROUND)
    if "positions" in want:
        configs["positions"] = positions_bench(args.docs, nq, 10, ROUND)
    if "out_of_cache" in want:
        # a shard whose .doc alone exceeds the 256 MiB Infinity Cache: here "fraction of HBM roofline" means HBM
        del shard
        big = Shard(args.big_docs, 0, 0, 1)
        oc = {"docs": args.big_docs, "doc_file_bytes": int(big.seg.doc_bytes.size), "index_build_s": round(big.build_s, 2),
              "segment_upload_s": round(big.upload_s, 3)}
        oc["cold"] = cold_bench(big, "%s_cold_big" % ROUND)
        oc["block_decode"] = decode_bench(big, 3, "%s_decode_big" % ROUND)
        # parity on the WHOLE batch for TERM and AND (the oracle does 1024 conjunctions over 100 M docs in a couple of seconds on the
        # GPU box's cores; the 10-clause OR over 100 M docs takes the oracle ~8 ms of one core per query: the whole batch fits too)
        oc["term"] = strip(search_config(big, "term", 5, 1, False, "%s_term_big" % ROUND, 0.0, nq, nq))
        oc["and3"] = strip(search_config(big, "and3", 3, 1, False, "%s_and3_big" % ROUND, 0.0, nq, nq))
        oc["or10"] = strip(search_config(big, "or10", 2, 1, False, "%s_or10_big" % ROUND, 0.0, nq, nq))
        configs["out_of_cache"] = oc
    if configs:
        out["configs"] = configs
    if args.workload == "and3":
        out["and3_queries_per_sec"] = out["queries_per_sec"]
        out["and3_roofline_frac"] = out["roofline"]["frac"]
    if world > 1 and "and3" in configs:
        out["and3_whole_index_queries_per_sec"] = configs["and3"]["queries_per_sec"]
        out["and3_whole_index_docs"] = world * args.docs
    if dist_mode:
        out["gathers_issued"] = comm.gathers_issued()

    # ---- the full tree -> a file + stderr; the ONE stdout line = the contract's keys + the north-star's scalars (final_line) ------
    full = _jsonable(out)
    detail_paths = [args.detail] if args.detail else [os.path.join(ROOT, "bench_detail.json")]
    if not args.detail and os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        detail_paths.append(os.path.join(ROOT, "gpurun_out", "bench_detail.json"))
    text = final_line(full, os.path.basename(detail_paths[0])) if rank == 0 else ""
    if rank == 0:
        for path in detail_paths:
            try:
                with open(path, "w") as f:
                    json.dump(full, f, indent=1)
            except OSError as e:
                print("could not write %s: %s" % (path, e), file=sys.stderr)
        print("bench detail (full tree; the contract line follows on stdout): " + json.dumps(full), file=sys.stderr, flush=True)
    sys.stdout.flush()
    os.dup2(stdout_fd, 1)
    os.close(stdout_fd)
    if rank == 0:
        print(text, flush=True)
    os.dup2(2, 1)  # anything native libraries print while shutting down stays off stdout too
    if dist_mode:
        dist.barrier()
        comm.close()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
