"""Summarise rocprofv3 --pmc output dirs: per-counter mean over dispatches of one kernel."""
import csv, glob, sys, collections
kern = sys.argv[1]
for d in sys.argv[2:]:
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                acc["_dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k in sorted(acc):
        v = acc[k]
        print(f"{d.split('/')[-1]:28s} {k:36s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
