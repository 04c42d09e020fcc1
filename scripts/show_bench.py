"""Condensed view of a bench.py JSON line. usage: show_bench.py <file>"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])


def rl(r):
    return {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items() if k in ("kernel", "kernel_ms", "frac", "achieved", "bytes_per_launch", "scan_equivalent_gbs", "traffic", "pruning")}


def cfg(name, c):
    keep = {k: c.get(k) for k in ("ms_per_step", "queries_per_sec", "postings_decoded_per_sec", "postings_covered_per_sec", "parity_vs_oracle", "gpu_over_cpu",
                                  "kernels_ms", "kernels_ms_total", "wall_ms_incl_host_planning", "hbm_bytes_held_per_doc_file_byte", "upload_s_pcie") if k in c}
    print(name, keep)
    if "parity" in c:
        print("   parity", c["parity"])
    if "roofline" in c:
        print("   roofline", rl(c["roofline"]))
    if "kernels_ms_isolated" in c:
        print("   kernels", {k: round(v, 4) for k, v in c["kernels_ms_isolated"].items()})


print({k: d.get(k) for k in ("value", "ms_per_step", "parity_vs_oracle_full_batch", "gpu_over_cpu", "postings_decoded_per_sec", "postings_covered_per_sec", "n_gpus")})
print("hoisted", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d.items() if k.startswith(("and3_", "or10_", "big_", "cold_", "block_decode_", "sloppy2_", "phrase2_", "one_stream", "whole_index"))})
print("sharded_overhead", d.get("sharded_overhead"))
print("streams", d.get("streams"))
print("roofline", rl(d["roofline"]))
print("parity", d.get("parity"))
print("cpu", d.get("cpu_baseline"))
for k, c in d.get("configs", {}).items():
    if k == "out_of_cache":
        print("out_of_cache", {x: c[x] for x in ("docs", "doc_file_bytes", "index_build_s", "segment_upload_s")})
        for kk in ("cold", "block_decode", "term", "and3", "or10"):
            if kk in c:
                cfg("  big." + kk, c[kk])
    elif k == "positions":
        print("positions", {x: c[x] for x in ("docs", "doc_file_bytes", "pos_file_bytes", "index_build_s")})
        cfg("  positions_decode", c["positions_decode"])
        print("   ", {x: c["positions_decode"].get(x) for x in ("positions", "positions_decoded_per_sec", "parity_vs_oracle")})
        if "sloppy2" in c:
            cfg("  sloppy2", c["sloppy2"])
        cfg("  phrase2", c["phrase2"])
        ph = c["phrase2"]
        print("   ", {"lead_postings_per_step": ph.get("lead_postings_per_step", ph.get("conjunction_matches_checked_per_step")),  # (its name before the round's last commits)
                      "phrase_hits_per_step": ph.get("phrase_hits_per_step"), "cpu_baseline": ph.get("cpu_baseline")})
    else:
        cfg(k, c)
