"""The JSON line bench.py prints (the committed line of the round's final run, profiles/r03_bench.json) carries every
field of the driver's contract, and its numbers are consistent with each other and with SURVEY.md 8d's byte counts."""
import json
import os

import numpy as np

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields_and_consistent_numbers():
    d = json.load(open(os.path.join(ROOT, "profiles", "r03_bench.json")))
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
    # the measured traffic is quoted for the build it was taken on, and says where it comes from
    assert "traffic_source" in r and "rocprofv3" in r["traffic_source"]
    # the caller side on BASELINE.json configs[1]: every problem solved, judged by the evaluator, a CPU figure beside it
    sv = d["other"]["solve_config2_B1024_N40"]
    assert sv["problems"] == 1024 and sv["solved_to_1e-6"] == 1024 and sv["violation_max"] <= 1e-6 * 1.0001
    assert sv["solved_problems_per_s"] > 1e3
    cs = sv["cpu_baseline"]
    assert cs["cores"] == 1 and cs["kind"] == "port" and cs["solved_to_1e-6"] == cs["problems"] >= 1 and "not Ipopt" in cs["sample"]
    # one measurement for every N: the timed region holds the K launches only, the end-of-job tail is reported beside it
    assert "K-launch region only" in d["config"]["timing"] and d["gather_ms"] > 0 and d["gather_c_ms"] > 0
    assert d["config"]["multi_gpu_status"] == "single GPU" and "generated on the device" in d["config"]["workload_data"]
    # the tracked rocprofv3 average of the newest profile round stands beside the live one (and a note when they differ by > 5 %)
    assert r["profile_launch_ms_avg"] > 0
    if abs(r["profile_launch_ms_avg"] - r["launch_ms_avg"]) > 0.05 * r["profile_launch_ms_avg"]:
        assert "profile_vs_live" in r


def test_kernel_build_id_is_stable_and_traffic_lookup_refuses_other_builds(tmp_path, monkeypatch):
    """bench.measured_traffic quotes profiles/traffic.json only for the kernel build it was measured on."""
    import json as J

    bid = bench.kernel_build_id()
    assert len(bid) == 16 and bid == bench.kernel_build_id()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / "profiles")
    os.makedirs(tmp_path / "quadruped_landing_amd" / "csrc")
    for f in ("qln_kernels.hip", "qln_kernel_common.h", "qln_device.h"):
        (tmp_path / "quadruped_landing_amd" / "csrc" / f).write_text("x" + f)
    here = bench.kernel_build_id()
    J.dump({"config3": {"hbm_bytes_per_launch": 7.0e9, "build_id": here, "profile": "p"},
            "config4": {"hbm_bytes_per_launch": 1.4e10, "build_id": "somethingelse"}}, open(tmp_path / "profiles" / "traffic.json", "w"))
    assert bench.measured_traffic("config3")[0] == 7.0e9
    t, why = bench.measured_traffic("config4")
    assert t is None and "not quoted" in why
    t, why = bench.measured_traffic("config2")
    assert t is None and "no PMC profile" in why
