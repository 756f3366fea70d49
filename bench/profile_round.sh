#!/bin/bash
# usage (on the GPU box, via gpurun):  bench/profile_round.sh r01 [workload] [dense_blocks|structural]
# 1. rocprofv3 --kernel-trace --stats of the default bench command  -> kernel_stats.csv
# 2. separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ set)          -> pmc_summary.txt, traffic.json
# Everything lands in gpurun_out/profile_<tag>/ (a fresh tag per round of profiling: nothing is overwritten);
# bench/collect_profiles.py then files every such directory under profiles/ as <round>_<key>_<n>_*, n counting up, and
# rewrites profiles/ROUNDS.md (one line per profile round) and profiles/traffic.json (the newest round per workload).
TAG=${1:-r01}; WL=${2:-config3}; FMT=${3:-dense_blocks}
KEY=$WL; [ "$FMT" = structural ] && KEY=${WL}_structural
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profile_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --workload $WL --jac-format $FMT --steps 20 --warmup 3 --placement-trials 1 --no-cpu-baseline --no-other > $OUT/bench_under_rocprof.json 2> $OUT/stats.log || { tail -5 $OUT/stats.log; exit 1; }
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
# 1b. the same command with the default setup (Jacobian buffer placed across two 32-GiB regions).  The setup times the
#     same kernel on ~120 windows, so rocprofv3's --stats average mixes those with the timed launches; the per-dispatch
#     trace of the run is therefore summarised over its LAST steps+warmup dispatches as well.
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_placed -- python3 $R/bench.py --workload $WL --jac-format $FMT --steps 20 --warmup 3 --no-cpu-baseline --no-other > $OUT/bench_under_rocprof_placed.json 2> $OUT/stats_placed.log || { tail -5 $OUT/stats_placed.log; exit 1; }
cp $(find $OUT/stats_placed -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_placed_all_dispatches.csv
python3 - <<PY
import csv, glob
rows = [r for f in glob.glob("$OUT/stats_placed/**/*kernel_trace.csv", recursive=True) for r in csv.DictReader(open(f)) if "k_constraint_jacobian" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-23:]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in last]
open("$OUT/kernel_stats_placed_timed_launches.csv", "w").write(
    '"Name","Calls","TotalDurationNs","AverageNs","MinNs","MaxNs"\n"%s (last %d of %d dispatches of the run: 3 warm-up + 20 timed)",%d,%d,%.1f,%d,%d\n'
    % (last[0]["Kernel_Name"], len(d), len(rows), len(d), sum(d), sum(d) / len(d), min(d), max(d)))
print(open("$OUT/kernel_stats_placed_timed_launches.csv").read())
PY
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_STALL_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_p$i -- python3 $R/bench.py --workload $WL --jac-format $FMT --steps 5 --warmup 1 --placement-trials 1 --no-cpu-baseline --no-other > $OUT/pmc_p$i.log 2>&1 || { echo "pmc pass $i failed"; tail -5 $OUT/pmc_p$i.log; exit 1; }
done
python3 $R/bench/pmc_summary.py k_constraint_jacobian $OUT/pmc_p* > $OUT/pmc_summary.txt
python3 - <<PY
import json,re
vals={}
for line in open("$OUT/pmc_summary.txt"):
    p=line.split()
    vals[p[1]]=float(p[-1].split("=")[1])
# gfx950: FETCH_SIZE (KiB) reads exactly 1/2 of the bytes of coalesced 8- and 16-B-per-lane loads
# (MI355X_MICROARCH.md HBM section; re-calibrated here with bench/store_ceiling.hip: profiles/calibration_*.txt);
# WRITE_SIZE (KiB) is exact.
fetch=2*vals["FETCH_SIZE"]*1024; write=vals["WRITE_SIZE"]*1024
import sys
sys.path.insert(0, "$R")
import bench
import csv
def avg_ms(path):
    try:
        r = [x for x in csv.DictReader(open(path)) if "k_constraint_jacobian" in x["Name"]]
        return float(r[0]["AverageNs"]) / 1e6
    except Exception:
        return None
def events_ms(path):
    try:
        return json.load(open(path))["roofline"]["launch_ms_avg"]
    except Exception:
        return None
json.dump({"$KEY":{"hbm_bytes_per_launch":fetch+write,"fetch_bytes_corrected":fetch,"write_bytes":write,
  "FETCH_SIZE_KiB_raw":vals["FETCH_SIZE"],"WRITE_SIZE_KiB_raw":vals["WRITE_SIZE"],
  "build_id":bench.kernel_build_id(),"profile":"profiles/${TAG}_pmc_summary_$KEY.txt",
  "kernel_avg_ms_rocprof_placed":avg_ms("$OUT/kernel_stats_placed_timed_launches.csv"),
  "kernel_avg_ms_hip_events_placed":events_ms("$OUT/bench_under_rocprof_placed.json"),
  "kernel_avg_ms_rocprof_plain":avg_ms("$OUT/kernel_stats.csv"),
  "kernel_avg_ms_hip_events_plain":events_ms("$OUT/bench_under_rocprof.json"),
  "note":"per launch of k_constraint_jacobian; fetch = 2 x FETCH_SIZE x 1024 (gfx950 correction), write = WRITE_SIZE x 1024"}},
  open("$OUT/traffic.json","w"),indent=1)
print(open("$OUT/traffic.json").read())
PY
cat $OUT/kernel_stats.csv | head -4
cat $OUT/bench_under_rocprof.json
