"""Split a loop of the 4-wave prefill kernel's ISA (hipcc -S) into MFMA slots and summarise every slot's fillers.
python tools/dbg/isa_slots.py file.s START_LINE END_LINE [-v]"""
import re, sys
path, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
verbose = "-v" in sys.argv
lines = open(path).read().split("\n")[a - 1:b]
slots, cur = [], []
skip_to = None
for l in lines:
    t = l.strip()
    if skip_to:  # inside a forward-branched-over (rare) block
        if t.startswith(skip_to + ":"):
            skip_to = None
        continue
    m = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", t)
    if m and any(x.strip().startswith(m.group(1) + ":") for x in lines[lines.index(l) + 1:]):
        cur.append(t.split(";")[0].strip())
        skip_to = m.group(1)
        continue
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
        continue
    t = t.split(";")[0].strip()
    if not t:
        continue
    cur.append(t)
    if t.startswith("v_mfma"):
        slots.append(cur)
        cur = []
slots.append(cur)
def cost(m, t):
    if m.startswith("v_mfma"): return 8
    if m.startswith(("v_exp", "v_rcp", "v_log", "v_rsq", "v_sqrt")): return 8
    if m == "s_nop": return 4 * 1 if int(t.split()[1]) < 4 else int(t.split()[1]) + 1
    return 4
tot = 0
for i, s in enumerate(slots):
    # a slot here = the fillers BEFORE the MFMA that ends it
    kinds = {}
    c = 0
    for t in s:
        m = t.split()[0]
        c += cost(m, t)
        key = m
        if m == "s_waitcnt": key = t.replace("s_waitcnt ", "W:")
        if m == "s_nop": key = "nop" + t.split()[1]
        kinds[key] = kinds.get(key, 0) + 1
    tot += c
    print(f"{i:3d} n={len(s):3d} issue~{c:4d}  " + " ".join(f"{k}x{v}" if v > 1 else k for k, v in kinds.items()))
    if verbose:
        for t in s: print("        ", t)
print("total instructions", sum(len(s) for s in slots), "issue estimate", tot)
