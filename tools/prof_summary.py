"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid, block) count / median / min / mean duration.

    python tools/prof_summary.py <dir-or-kernel_trace.csv> [substring filter] [--json]
"""
import csv
import glob
import os
import re
import statistics
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\(.*", "", name)  # drop argument list
    name = name.replace("void ", "").replace("cvllm::", "")
    return name[:110]


def main():
    as_json = "--json" in sys.argv
    argv = [a for a in sys.argv if a != "--json"]
    path = argv[1]
    filt = argv[2] if len(argv) > 2 else ""
    files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
    rows = defaultdict(list)
    for f in files:
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if filt and filt not in r["Kernel_Name"]:
                    continue
                key = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1),
                       int(r["Workgroup_Size_X"]), r["VGPR_Count"], r["LDS_Block_Size"])
                rows[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    if as_json:
        import json

        out = {}
        for key, d in rows.items():
            e = out.setdefault(key[0], {"n": 0, "sum_ns": 0})
            e["n"] += len(d)
            e["sum_ns"] += sum(d)
        print(json.dumps({k: {"n": v["n"], "mean_us": round(v["sum_ns"] / v["n"] / 1e3, 2)} for k, v in out.items()}, indent=1))
        return
    print(f"{'kernel':110s} {'wgs':>7s} {'blk':>5s} {'vgpr':>5s} {'lds':>7s} {'n':>5s} {'med_us':>9s} {'min_us':>9s} {'mean_us':>9s}")
    for key, d in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        print(f"{key[0]:110s} {key[1]:7d} {key[2]:5d} {key[3]:>5s} {key[4]:>7s} {len(d):5d} "
              f"{statistics.median(d) / 1e3:9.2f} {min(d) / 1e3:9.2f} {statistics.mean(d) / 1e3:9.2f}")


if __name__ == "__main__":
    main()
