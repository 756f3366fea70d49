"""File the profile rounds bench/profile_round.sh left under gpurun_out/profile_<tag>/ into profiles/ WITHOUT overwriting
earlier ones: every round becomes profiles/<round>_<key>_<n>_{kernel_stats_*.csv, bench_*.json, pmc_summary.txt}, n counting
up per (round, key); profiles/rounds.jsonl gets one record per profile round (append-only), profiles/ROUNDS.md is rewritten
from it (per-round kernel averages: the tracked evidence brackets whatever box the driver lands on), and profiles/traffic.json
keeps the NEWEST round of every workload key (what bench.py quotes, stamped with the kernel build).
    python bench/collect_profiles.py r03 [tag ...]        (default: every gpurun_out/profile_* not filed yet)"""
import glob
import json
import os
import shutil
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")
rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
tags = sys.argv[2:] or [os.path.basename(d)[len("profile_"):] for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"profile_{rnd}*")))
                        if os.path.isdir(d)]  # by default only this round's tags (profile_round.sh <round><suffix> ...)
log_path = os.path.join(PROF, "rounds.jsonl")
done = set()
if os.path.exists(log_path):
    done = {json.loads(l)["source"] for l in open(log_path) if l.strip()}
traffic_path = os.path.join(PROF, "traffic.json")
traffic = json.load(open(traffic_path)) if os.path.exists(traffic_path) else {}
FILES = ["kernel_stats.csv", "kernel_stats_placed_timed_launches.csv", "kernel_stats_placed_all_dispatches.csv", "bench_under_rocprof.json",
         "bench_under_rocprof_placed.json", "pmc_summary.txt"]
for tag in tags:
    src = os.path.join(ROOT, "gpurun_out", f"profile_{tag}")
    tj = os.path.join(src, "traffic.json")
    if not os.path.exists(tj):
        print(f"{tag}: no traffic.json (incomplete round), skipped")
        continue
    entry = json.load(open(tj))
    (key, e), = entry.items()
    ident = f"{tag}:{e.get('build_id')}:{e.get('hbm_bytes_per_launch')}"
    if ident in done:
        print(f"{tag}: already filed")
        continue
    n = 1
    while glob.glob(os.path.join(PROF, f"{rnd}_{key}_{n}_*")):
        n += 1
    prefix = f"{rnd}_{key}_{n}"
    for f in FILES:
        if os.path.exists(os.path.join(src, f)):
            shutil.copy(os.path.join(src, f), os.path.join(PROF, f"{prefix}_{f}"))
    e["profile"] = f"profiles/{prefix}_pmc_summary.txt"
    traffic[key] = e
    rec = {"source": ident, "round": rnd, "key": key, "n": n, "tag": tag, "filed": time.strftime("%Y-%m-%d %H:%M"), **e}
    with open(log_path, "a") as fh:
        fh.write(json.dumps(rec) + "\n")
    print(f"{tag}: filed as profiles/{prefix}_*")
json.dump(traffic, open(traffic_path, "w"), indent=1, sort_keys=True)

recs = [json.loads(l) for l in open(log_path) if l.strip()] if os.path.exists(log_path) else []
ALG = {"config3": 6950486016.0, "config3_structural": 2069364736.0}
with open(os.path.join(PROF, "ROUNDS.md"), "w") as fh:
    fh.write("# Profile rounds of the hot kernel (`bench/profile_round.sh`), one line per round, none overwritten\n\n"
             "`rocprofv3 --kernel-trace --stats` average of `k_constraint_jacobian` over the run's last 23 dispatches (3 warm-up + 20 timed) "
             "beside the HIP-event average `bench.py` measured in the same run, for the region-placed buffer (the default) and for a plain "
             "allocation; PMC traffic per launch from the separate `--pmc` passes of the same round.  Every round is a different box.\n\n"
             "| files `profiles/<prefix>_*` | workload | kernel build | placed: rocprofv3 / HIP events (ms) | plain: rocprofv3 / HIP events (ms) | HBM traffic per launch (GB) |\n|---|---|---|---|---|---|\n")
    f4 = lambda v: "–" if v is None else f"{v:.4f}"
    gaps = []
    for r in recs:
        a, b = r.get('kernel_avg_ms_rocprof_placed'), r.get('kernel_avg_ms_hip_events_placed')
        if a and b and b > 1.05 * a:
            gaps.append(f"`{r['round']}_{r['key']}_{r['n']}` (placed run: {b:.4f} ms by HIP events against {a:.4f} ms per kernel)")
        fh.write(f"| `{r['round']}_{r['key']}_{r['n']}` | {r['key']} | `{r.get('build_id')}` | {f4(r.get('kernel_avg_ms_rocprof_placed'))} / "
                 f"{f4(r.get('kernel_avg_ms_hip_events_placed'))} | {f4(r.get('kernel_avg_ms_rocprof_plain'))} / {f4(r.get('kernel_avg_ms_hip_events_plain'))} | "
                 f"{r['hbm_bytes_per_launch'] / 1e9:.3f} |\n")
    if gaps:
        fh.write("\nSince round 3's `bench.py` the HIP-event figure is ONE pair of events around the K back-to-back launches of the timed region, / K: it "
                 "contains whatever gaps the host leaves between launches.  Un-profiled there are none (the figure equals the per-kernel average); under "
                 "`rocprofv3` the host can fall behind its queue, and the figure then exceeds the per-kernel average of the very same dispatches -- a "
                 "property of the profiled run, not of the kernel: " + "; ".join(gaps) + ".\n")
print(open(os.path.join(PROF, "ROUNDS.md")).read())
