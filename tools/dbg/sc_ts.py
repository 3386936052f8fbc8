"""Phase stamps of the SnapKV kernels (debug library built with -DCVLLM_SC_TS):  python tools/dbg/sc_ts.py <lib.so> [nwg]"""
import ctypes, os, sys
import numpy as np
os.environ["CVLLM_LIB_PATH"] = os.path.abspath(sys.argv[1])
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "compactor-vllm_amd"))
import torch
from compactor_vllm_amd import _lib
from compactor_vllm_amd.compression.snapkv import query_aware_key_scores
L = 32768
torch.manual_seed(0)
q = torch.randn(L, 32, 128, device="cuda", dtype=torch.bfloat16)
k = torch.randn(L, 8, 128, device="cuda", dtype=torch.bfloat16)
cu = torch.tensor([0, L], device="cuda", dtype=torch.int32)
for _ in range(5):
    query_aware_key_scores(q, k, cu, cu, w=32, max_seqlen_k=L)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (1024 * 16))()
_lib.lib().cvllm_debug_scoring_stamps(buf)
a = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 16)[:512].astype(np.int64)
for base, name in ((0, "pass 1"), (8, "pass 2")):
    t = a[:, base:base + 8]
    t0 = t[:, 0].min()
    rel = (t - t0) / 100.0  # us
    print(name, "stamps (us after the first workgroup's entry): entry, scalars, first tile staged, tile 0..3 done, exit")
    print("  median", np.round(np.median(rel, axis=0), 2))
    print("  min   ", np.round(rel.min(axis=0), 2))
    print("  max   ", np.round(rel.max(axis=0), 2))
