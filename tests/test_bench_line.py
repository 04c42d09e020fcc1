"""bench.py's ONE stdout line (the driver parses the last line of stdout; round 5's 25 KB line came back `parsed: null`):
final_line() must keep every contract key, stay under its byte cap whatever the tree holds, and round-trip as strict JSON."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _tree():
    """a full result tree as a real run produced it (round 5's, kept under profiles/), with this round's additions"""
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_n1.json")))
    full["ms_per_step_min_median_max"] = [0.051, 0.0533, 0.0611]
    full["latency_batch_of_one"] = {k: {"p50_us": 61.25, "p99_us": 140.123456, "planned_p50_us": 80.0, "planned_p99_us": 190.0, "same_as_batch": True}
                                    for k in ("term", "and3", "or10")}
    full["latency_batch_of_one"]["note"] = "x" * 500
    full["config"]["timed_regions"] = "5 regions of 20 steps"
    return full


def test_line_has_the_contract_keys_and_fits():
    full = _tree()
    assert len(json.dumps(full)) > 20_000   # the tree itself is what the driver could not parse
    text = bench.final_line(full)
    assert "\n" not in text and len(text.encode()) < bench.LINE_MAX_BYTES <= 8192
    line = json.loads(text)
    for key in bench.LINE_REQUIRED + ("cpu_baseline", "parity"):
        assert key in line, key
    assert line["metric"] == full["metric"] and line["n_gpus"] == 1 and line["higher_is_better"] is True and line["scaling"] == "weak"
    assert line["config"]["workload"] == full["config"]["workload"] and "model" not in line["config"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in line["roofline"], key
    assert line["roofline"]["bound"] == "hbm" and 0 < line["roofline"]["frac"] <= 1 and line["roofline"]["peak"] == 8000.0
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in line["cpu_baseline"], key
    assert line["cpu_baseline"]["kind"] in ("port", "reference")
    assert abs(line["value"] - full["value"]) <= 1e-4 * full["value"]
    # the north-star's other targets travel as scalars
    for key in ("and3_queries_per_sec", "and3_roofline_frac", "block_decode_frac", "big_and3_roofline_frac", "or10_queries_per_sec"):
        assert isinstance(line[key], (int, float)), key
    assert line["latency_us_p50_p99"]["term"] == [61.25, 140.12]
    assert all(not isinstance(v, dict) or k in ("config", "roofline", "cpu_baseline", "parity", "latency_us_p50_p99", "latency_planned_us_p50_p99")
               for k, v in line.items())


def test_line_sheds_hoisted_scalars_before_it_grows_past_the_cap():
    full = _tree()
    full["config"]["workload"] = "w" * 100
    full["cpu_baseline"]["sample"] = "s" * 5000      # clipped
    full["roofline"]["bytes_are"] = "b" * 5000       # clipped
    text = bench.final_line(full)
    assert len(text.encode()) < bench.LINE_MAX_BYTES
    line = json.loads(text)
    assert len(line["cpu_baseline"]["sample"]) <= 240 and len(line["roofline"]["bytes_are"]) <= 200
    # a tree that would not fit even so: hoisted names (here: made absurdly long values impossible, so shrink the cap instead)
    old = bench.LINE_MAX_BYTES
    try:
        bench.LINE_MAX_BYTES = 2600
        small = json.loads(bench.final_line(full))
        assert small.get("truncated") is True
        for key in bench.LINE_REQUIRED:
            assert key in small
    finally:
        bench.LINE_MAX_BYTES = old


def test_line_refuses_a_tree_without_the_contract_keys():
    full = _tree()
    del full["roofline"]
    with pytest.raises(AssertionError):
        bench.final_line(full)
    full = _tree()
    full["value"] = float("nan")
    with pytest.raises(ValueError):
        bench.final_line(full)


def test_jsonable_drops_private_keys_and_numpy_types():
    import numpy as np
    tree = {"a": np.float32(1.5), "b": np.arange(3), "_rows": np.zeros(4), "c": {"_x": 1, "y": float("inf")}, 7: (np.int64(2),)}
    assert bench._jsonable(tree) == {"a": 1.5, "b": [0, 1, 2], "c": {"y": None}, "7": [2]}
