"""Average SQ counters per launch of one kernel from a rocprofv3 --pmc run.   python tools/pmc_sq.py <dir> <kernel substring>"""
import csv, glob, os, statistics, sys
d, filt = sys.argv[1], sys.argv[2]
vals = {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            if filt in r["Kernel_Name"]:
                vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k in sorted(vals):
    print(f"{k:34s} n={len(vals[k]):4d} mean={statistics.mean(vals[k]):16.0f}")
