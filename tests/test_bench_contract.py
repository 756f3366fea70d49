"""The JSON line bench.py prints (the committed line of the round's final run, profiles/r01_bench.json) carries every
field of the driver's contract, and its numbers are consistent with each other and with SURVEY.md 8d's byte counts."""
import json
import os

import numpy as np

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields_and_consistent_numbers():
    d = json.load(open(os.path.join(ROOT, "profiles", "r01_bench.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"].startswith("knot-point constraint+Jacobian evals/sec") and d["unit"] == "knot-evals/s"
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    B, N = d["config"]["problems_per_gpu"], d["config"]["knots"]
    # value = units of all ranks / wall time per step
    assert abs(d["value"] - B * N / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    alg = float(np.sum(bench.algorithmic_bytes(N, np.full(B, 14))))
    assert abs(r["algorithmic_bytes_per_launch"] - alg) <= 1.0 and abs(alg / (B * N) - 2651.4) < 1e-9
    assert abs(r["achieved"] - alg / (r["launch_ms_avg"] * 1e-3) / 1e9) <= 1e-6 * r["achieved"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-12
    assert 0.99 * alg <= r["traffic"] <= 1.05 * alg          # PMC traffic: no wasted re-reads
    assert r["launch_ms_avg"] <= d["ms_per_step"]            # the kernel inside the wall-clock step
    s = r["strict_nnz"]
    assert abs(s["bytes_per_knot_eval"] - float(bench.strict_bytes(N, 14)) / N) < 1e-9
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0
    # north star: >= 1e7 knot-evals/s at >= 40 % of the HBM roofline
    assert d["value"] >= 1e7 and r["frac"] >= 0.40
