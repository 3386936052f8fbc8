"""In-kernel phase timestamps of decode_fused_kernel (s_memtime, workgroup 0 / wave 0) from a debug build.

    python tools/decode_ts.py build     # here: hipcc -DCVLLM_DEC_TS -> tools/dbg/libcvllm_hip_ts.so
    python tools/decode_ts.py           # on the GPU box
"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DBG = os.path.join(ROOT, "tools", "dbg", "libcvllm_hip_ts.so")
CS = os.path.join(ROOT, "compactor-vllm_amd", "csrc")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    os.makedirs(os.path.dirname(DBG), exist_ok=True)
    srcs = [os.path.join(CS, f) for f in sorted(os.listdir(CS)) if f.endswith(".hip")]
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DCVLLM_DEC_TS",
           "-mllvm", "-amdgpu-kernarg-preload-count=16", *srcs, "-o", DBG]
    subprocess.check_call(cmd)
    print(DBG)
    sys.exit(0)
sys.path.insert(0, os.path.join(ROOT, "compactor-vllm_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from compactor_vllm_amd import _lib
_lib.LIB_PATH = DBG
import compactor_vllm_amd.attention.sparse_decode_kernel as dk
from microbench import build_cache
dev = torch.device("cuda:0")
names = ["args", "bmap", "len", "loads retired", "ring primed", "loop done", "wave merged", "barrier", "end"]
for L, S in [(16, 32), (16384, 32)]:
    caches, pt, bm, lens = build_cache(1, 8, 128, 128, max(L, 128), torch.bfloat16, dev, 4)
    lens.fill_(L)
    q = torch.randn(1, 32, 128, device=dev, dtype=torch.bfloat16)
    dk.plan_internal_splits = lambda n_bh, bound, ks, S=S: S
    rows = []
    for it in range(12):
        for kc, vc in caches:
            dk.head_sparse_decode_attention(q, kc, vc, lens, pt, bm, 8, 128)
            torch.cuda.synchronize()
            ts = (ctypes.c_ulonglong * 16)()
            _lib.lib().cvllm_debug_read_decode_ts(ts)
            rows.append([ts[i] - ts[0] for i in range(9)])
    rows = rows[8:]
    med = [sorted(r[i] for r in rows)[len(rows) // 2] for i in range(9)]
    print(f"L={L} S={S}: cycles since kernel entry (s_memtime ticks, 100 MHz => x10 ns)")
    for n, v in zip(names, med):
        print(f"   {n:14s} {v:8d}")
