"""Phase stamps of the leverage / chunk-mass kernels (debug library with -DCVLLM_SC_TS):  python tools/dbg/sc_ts2.py <lib.so>"""
import ctypes, os, sys
import numpy as np
os.environ["CVLLM_LIB_PATH"] = os.path.abspath(sys.argv[1])
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "compactor-vllm_amd"))
import torch
from compactor_vllm_amd import _lib
from compactor_vllm_amd.compression import compactor as C
L, HQ, HKV, D = 32768, 32, 8, 128
torch.manual_seed(0)
q = torch.randn(L, HQ, D, device="cuda", dtype=torch.bfloat16)
k = torch.randn(L, HKV, D, device="cuda", dtype=torch.bfloat16)
cu = torch.tensor([0, L], device="cuda", dtype=torch.int32)
phi = torch.randn(D, 48, device="cuda", dtype=torch.bfloat16)


def stamps(n, nwg):
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (1024 * 16))()
    _lib.lib().cvllm_debug_scoring_stamps(buf)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 16)[:nwg, :n].astype(np.int64)
    rel = (a - a[:, 0].min()) / 100.0
    for nm, f in (("median", np.median), ("min", np.min), ("max", np.max)):
        print("  %-6s" % nm, np.round(f(rel, axis=0), 2))


from compactor_vllm_amd.compression.compactor import approximate_leverage_scores, non_causal_attn_scores
q = (q.float() * 0.3).to(torch.bfloat16); k = (k.float() * 0.3).to(torch.bfloat16)
v = torch.randn(L, HKV, D, device="cuda", dtype=torch.bfloat16)
PHI = (torch.randn(D, 48, device="cuda") / 48 ** 0.5).to(torch.bfloat16)
for _ in range(4):
    pre = approximate_leverage_scores(k, [L], PHI, normalize=True, chunk_size=512)
print("leverage_fused2_kernel, 512 workgroups: load issued, PHI staged, sketch done, means done, Gram reduced, W ready, exit")
stamps(7, 512)
for _ in range(4):
    non_causal_attn_scores(q, k, v, cu, L, chunk_size=128, sm_scale=1.0, normalize=True)
print("chunk_mass_kernel, first 1024 of 2048 workgroups: entry, K tile staged, query block 0..3 started, loop done, exit")
stamps(8, 1024)
