"""Per-slot timeline of one wave of the 4-wave prefill kernel (debug builds with -DP4_TS_SLOT=k, see prefill_attn.hip).

  python tools/dbg/slot_ts.py build [EXTRA -D flags ...]   # here or on the box: 65 small libraries, one per stamp position
  python tools/dbg/slot_ts.py run [L] [wave]               # on the GPU box: run each, print the timeline

Every library holds only the BF16 / G = 4 instantiation.  A run launches the kernel a few times at 1 x L tokens and
reads, for every workgroup that has a steady-state tile 16, s_memtime at the tile's entry, in front of MFMA slot k and
behind the tile's barrier; the table is the median over workgroups of (slot k - entry), slot by slot.
"""
import ctypes
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
SRC = os.path.join(ROOT, "compactor-vllm_amd", "csrc", "prefill_attn.hip")
OUT = os.environ.get("SLOT_DIR", os.path.join(ROOT, "tools", "dbg", "variants", "slot"))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-honor-nans", "-mllvm", "-amdgpu-mfma-vgpr-form",
         "-fno-slp-vectorize", "-shared"]


def build(extra):
    os.makedirs(OUT, exist_ok=True)

    def one(k):
        cmd = ["/opt/rocm/bin/hipcc", *FLAGS, f"-DP4_TS_SLOT={k}", *extra, SRC, "-o", os.path.join(OUT, f"libpf_slot{k}.so")]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(r.stderr[-2000:])

    with ThreadPoolExecutor(max_workers=8) as ex:
        list(ex.map(one, range(65)))
    print("built 65 libraries in", OUT)


def run(L, wave):
    import numpy as np
    import torch

    HQ, HKV, D, PS = 32, 8, 128, 128
    dev = "cuda"
    torch.manual_seed(0)
    q = torch.randn(L, HQ, D, device=dev, dtype=torch.bfloat16)
    k = torch.randn(L, HKV, D, device=dev, dtype=torch.bfloat16)
    v = torch.randn(L, HKV, D, device=dev, dtype=torch.bfloat16)
    out = torch.empty_like(q)
    kc = torch.zeros(PS * 16, D, device=dev, dtype=torch.bfloat16)
    vc = torch.zeros_like(kc)
    pt = torch.zeros(2, HKV, 4, dtype=torch.int32, device=dev)
    lens = torch.zeros(1, HKV, dtype=torch.int32, device=dev)
    bm = torch.ones(1, dtype=torch.int32, device=dev)
    cu = torch.tensor([0, L], dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    P, I, I64, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
    rows = []
    for kslot in range(65):
        lib = ctypes.CDLL(os.path.join(OUT, f"libpf_slot{kslot}.so"))
        fn = lib.cvllm_prefill_attn
        fn.restype = I
        fn.argtypes = [P, P, P, I64, I64, I64, I64, I64, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, F, I, P]
        for _ in range(6):
            st = fn(q.data_ptr(), k.data_ptr(), v.data_ptr(), HQ * D, HKV * D, D, HKV * D, D, kc.data_ptr(), vc.data_ptr(),
                    out.data_ptr(), lens.data_ptr(), pt.data_ptr(), bm.data_ptr(), cu.data_ptr(), 1, L, L, HQ, HKV, D, PS, 4,
                    D ** -0.5, 1, torch.cuda.current_stream().cuda_stream)
            assert st == 0, st
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * (8192 * 4))()
        lib.cvllm_debug_prefill_slot_stamps(buf)
        a = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 4).astype(np.int64)
        a = a[a[:, 3] > 0]
        dk = a[:, 1] - a[:, 0]
        tot = a[:, 2] - a[:, 0]
        rows.append((kslot, len(a), float(np.median(dk)), float(np.median(tot)), float(np.percentile(dk, 10)),
                     float(np.percentile(dk, 90))))
    print(f"# 4-wave prefill kernel, 1 x {L} tokens, wave {wave}: s_memtime (shader clock) from the entry of tile 16 to the front")
    print("# of MFMA slot k (0-15 phase A / 16-31 phase B of the tile's first unit, 32-63 second unit, 64 = before the barrier);")
    print("# median over the workgroups that have such a tile; 'step' = distance to the previous slot")
    print("slot  n_wg  median   step   p10   p90   tile_total")
    prev = 0.0
    for kslot, n, med, tot, p10, p90 in rows:
        print(f"{kslot:4d} {n:5d} {med:7.0f} {med - prev:6.0f} {p10:6.0f} {p90:6.0f} {tot:8.0f}")
        prev = med
    tt = np.median([r[3] for r in rows])
    print(f"# tile total (entry -> behind the barrier), median over builds: {tt:.0f} cycles = {tt / 64:.1f} per MFMA slot")


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 32768, int(sys.argv[3]) if len(sys.argv) > 3 else 0)
