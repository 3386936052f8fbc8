"""Per-workgroup phase stamps of decode_fused_kernel on ONE clock (s_memrealtime, 100 MHz = 10 ns) from the debug build
(tools/decode_ts.py build): where the launch's time goes across all workgroups - start skew, streaming, the in-launch
merge hand-off.

    python tools/decode_ts.py build     # here
    python tools/decode_rt.py [--L 16640] [--B 1] [--splits 32]     # on the GPU box
"""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DBG = os.environ.get("CVLLM_DBG_LIB") or os.path.join(ROOT, "tools", "dbg", "libcvllm_hip_ts.so")
sys.path.insert(0, os.path.join(ROOT, "compactor-vllm_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from compactor_vllm_amd import _lib
_lib.LIB_PATH = DBG
import compactor_vllm_amd.attention.sparse_decode_kernel as dk
from microbench import build_cache

ap = argparse.ArgumentParser()
ap.add_argument("--L", type=int, default=16640)
ap.add_argument("--B", type=int, default=1)
ap.add_argument("--splits", type=int, default=32)
a = ap.parse_args()
dev = torch.device("cuda:0")
names = ["entry", "metadata done", "ring issued", "loop done", "partials staged", "values published", "ML published",
         "mailbox full", "output stored", "exit", "all waves polled", "max ready", "weighted ready"]
ORDER = [0, 1, 2, 3, 4, 5, 6, 7, 10, 11, 12, 8, 9]
HKV = 8
caches, pt, bm, lens = build_cache(a.B, HKV, 128, 128, max(a.L, 128), torch.bfloat16, dev, 6)
lens.fill_(a.L)
q = torch.randn(a.B, 32, 128, device=dev, dtype=torch.bfloat16)
dk.plan_internal_splits = lambda n_bh, bound, ks, S=a.splits: S
nwg = a.B * HKV * a.splits
L = _lib.lib()
L.cvllm_debug_read_decode_rt.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_ulonglong * (1024 * 16))()
runs = []
ap_b2b = os.environ.get("RT_BACK_TO_BACK", "1") == "1"
for it in range(10):
    for ci in range(len(caches)):
        if ap_b2b:  # the stamped launch is the LAST of four queued back to back (no idle GPU in front of it)
            for j in range(3):
                kc, vc = caches[(ci + 1 + j) % len(caches)]
                dk.head_sparse_decode_attention(q, kc, vc, lens, pt, bm, HKV, 128)
        kc, vc = caches[ci]
        dk.head_sparse_decode_attention(q, kc, vc, lens, pt, bm, HKV, 128)
        torch.cuda.synchronize()
        L.cvllm_debug_read_decode_rt(buf)
        rows = [[buf[w * 16 + i] for i in range(13)] for w in range(min(nwg, 1024))]
        t0 = min(r[0] for r in rows)
        runs.append([[x - t0 for x in r] for r in rows])
runs = runs[12:]
print(f"B={a.B} L={a.L} splits={a.splits} workgroups={nwg}: microseconds since the FIRST workgroup's entry; per stamp the "
      f"min / median / max over workgroups, median over {len(runs)} launches")
for i in ORDER:
    n = names[i]
    mins = sorted(min(r[i] for r in run) for run in runs)[len(runs) // 2]
    meds = sorted(sorted(r[i] for r in run)[len(run) // 2] for run in runs)[len(runs) // 2]
    maxs = sorted(max(r[i] for r in run) for run in runs)[len(runs) // 2]
    print(f"   {n:18s} {mins / 100:7.2f} {meds / 100:7.2f} {maxs / 100:7.2f}")
# per-workgroup durations of the phases (median over workgroups and launches)
print("phase durations per workgroup (median / max over workgroups, median over launches), us:")
for a_, b_ in zip(ORDER[:-1], ORDER[1:]):
    d_med = sorted(sorted(r[b_] - r[a_] for r in run)[len(run) // 2] for run in runs)[len(runs) // 2]
    d_max = sorted(max(r[b_] - r[a_] for r in run) for run in runs)[len(runs) // 2]
    print(f"   {names[a_]:18s} -> {names[b_]:18s} {d_med / 100:7.2f} {d_max / 100:7.2f}")

# distribution of the entry and exit stamps of the LAST launch (sorted, every 16th workgroup), us
last = runs[-1]
for i in (0, 3, 9):
    v = sorted(r[i] for r in last if r[i] > -1e6)
    print(f"sorted '{names[i]}' stamps:", " ".join(f"{x / 100:.2f}" for x in v[::16]), f"... {v[-1] / 100:.2f}")
print("entry stamp by workgroup id (first 16):", " ".join(f"{r[0] / 100:.2f}" for r in last[:16]))
for x in range(8):
    grp = [r for i, r in enumerate(last) if i % 8 == x and r[3] > -1e6]
    print(f"  workgroups with id % 8 == {x}: entry {min(r[0] for r in grp) / 100:.2f}..{max(r[0] for r in grp) / 100:.2f}  "
          f"loop done {min(r[3] for r in grp) / 100:.2f}..{max(r[3] for r in grp) / 100:.2f}  exit {max(r[9] for r in grp) / 100:.2f}")
