"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into HBM bytes per launch (MI355X guide: units are
KiB; FETCH_SIZE under-counts wide coalesced reads by exactly 2x on gfx950 -> doubled).

    python tools/pmc_summary.py <fetch_dir> <write_dir> [kernel substring] [--command CMD --workload W --alg-bytes N]

With --command the output is the record bench.py reads back as `roofline.traffic` (profiles/rNN_bench_pmc.json).
"""
import csv, glob, json, os, statistics, sys


def collect(d, counter, filt):
    vals = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] != counter or filt not in r["Kernel_Name"]:
                    continue
                name = r["Kernel_Name"].split("(")[0].replace("void ", "")
                vals.setdefault(name, []).append(float(r["Counter_Value"]))
    return vals


def main():
    argv, meta = [], {}
    it = iter(sys.argv[1:])
    for a in it:
        if a in ("--command", "--workload", "--alg-bytes"):
            meta[a[2:]] = next(it)
        else:
            argv.append(a)
    fd, wd = argv[0], argv[1]
    filt = argv[2] if len(argv) > 2 else "cvllm::"
    F, W = collect(fd, "FETCH_SIZE", filt), collect(wd, "WRITE_SIZE", filt)
    out = {}
    for k in sorted(set(F) | set(W)):
        f = statistics.mean(F.get(k, [0.0]))
        w = statistics.mean(W.get(k, [0.0]))
        out[k] = {"launches": len(F.get(k, [])), "FETCH_SIZE_KiB_avg": round(f, 1), "WRITE_SIZE_KiB_avg": round(w, 1),
                  "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
    if "command" in meta:
        out = {"workload": meta.get("workload", "C3"),
               "command": meta["command"],
               "pmc_source": "separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes over the command, collection "
                             "limited to the roofline kernel (--kernel-include-regex); hbm bytes = (2 x FETCH_SIZE + "
                             "WRITE_SIZE) x 1024 (gfx950 rule of MI355X_MICROARCH.md)",
               "algorithmic_bytes_per_launch": int(meta.get("alg-bytes", 0)), "kernels": out}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
