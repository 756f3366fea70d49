"""Per-dispatch view of a rocprofv3 --pmc output dir: duration and counters of every dispatch of one kernel (used to
correlate the placement-dependent launch time with address-translation counters)."""
import csv, glob, sys, collections
kern, d = sys.argv[1], sys.argv[2]
rows = collections.OrderedDict()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            k = int(r["Dispatch_Id"])
            rows.setdefault(k, {"us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
            rows[k][r["Counter_Name"]] = float(r["Counter_Value"])
names = sorted({n for v in rows.values() for n in v if n != "us"})
print("dispatch  us      " + "  ".join(names))
for k, v in rows.items():
    print("%6d  %7.1f  " % (k, v["us"]) + "  ".join("%.4g" % v.get(n, float("nan")) for n in names))
