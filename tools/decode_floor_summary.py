"""Per-config kernel durations for tools/decode_floor.py run under rocprofv3 --kernel-trace (launch order, 40 per config)."""
import csv, glob, os, statistics, sys
files = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            if "decode_fused_kernel" in r["Kernel_Name"] or "decode_stage2" in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), "s2" if "stage2" in r["Kernel_Name"] else "s1",
                             int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])))
rows.sort()
cfgs = [(16, 32), (64, 32), (1024, 32), (2048, 32), (4096, 32), (8192, 32), (16384, 32), (16, 1), (16, 8), (2048, 8)]
s1 = [r for r in rows if r[1] == "s1"]
s2 = [r for r in rows if r[1] == "s2"]
i2 = 0
for ci, (L, S) in enumerate(cfgs):
    d = [r[2] for r in s1[ci * 40:(ci + 1) * 40]][8:]
    print(f"L={L:6d} S={S:3d} wgs={s1[ci*40][3]:4d} stage1 med {statistics.median(d)/1e3:7.2f} min {min(d)/1e3:7.2f} us")
if s2:
    d = [r[2] for r in s2]
    print(f"stage2 n={len(d)} med {statistics.median(d)/1e3:.2f} min {min(d)/1e3:.2f}")
