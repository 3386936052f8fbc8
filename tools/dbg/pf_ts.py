"""Per-workgroup start / loop / end stamps of the 4-wave prefill kernel (debug library with -DCVLLM_PF_TS):
python tools/dbg/pf_ts.py <lib.so> [L]"""
import ctypes, os, sys
import numpy as np
os.environ["CVLLM_LIB_PATH"] = os.path.abspath(sys.argv[1])
L = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "compactor-vllm_amd"))
import torch
from compactor_vllm_amd import _lib
from compactor_vllm_amd.attention.sparse_varlen_kernel import causal_sparse_varlen_with_cache
HQ, HKV, D, PS = 32, 8, 128, 128
dev = "cuda"
torch.manual_seed(0)
q = torch.randn(L, HQ, D, device=dev, dtype=torch.bfloat16)
k = torch.randn(L, HKV, D, device=dev, dtype=torch.bfloat16)
v = torch.randn(L, HKV, D, device=dev, dtype=torch.bfloat16)
kc = torch.zeros(PS * 16, D, device=dev, dtype=torch.bfloat16); vc = torch.zeros_like(kc)
pt = torch.zeros(2, HKV, 4, dtype=torch.int32, device=dev)
lens = torch.zeros(1, HKV, dtype=torch.int32, device=dev)
bm = torch.ones(1, dtype=torch.int32, device=dev)
cu = torch.tensor([0, L], dtype=torch.int32, device=dev)
for _ in range(12):
    causal_sparse_varlen_with_cache(q, k, v, kc, vc, lens, pt, bm, cu, L, 0, HKV, PS)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (8192 * 8))()
_lib.lib().cvllm_debug_prefill_stamps(buf)
nwg = (L // 64) * HKV
raw = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8)[:min(nwg, 8192)].astype(np.int64)
a = raw[:, :4]
clk = raw[:, 4:]
t0 = a[:, 0].min()
rel = (a - t0) / 100.0
bid = np.arange(len(a))
nqt = L // 64
qt = nqt - 1 - bid // HKV
ntiles = qt + 1  # 64-key tiles of the causal block (no cached prefix)
dur = rel[:, 3] - rel[:, 0]
pro = rel[:, 1] - rel[:, 0]
loop = rel[:, 2] - rel[:, 1]
epi = rel[:, 3] - rel[:, 2]
A = np.vstack([ntiles, np.ones_like(ntiles)]).T
coef = np.linalg.lstsq(A, loop, rcond=None)[0]
print(f"L={L}: {len(a)} workgroups; kernel span {rel[:,3].max():.1f} us")
print(f"  prologue (entry -> loop)  median {np.median(pro):.2f} us  max {pro.max():.2f}")
print(f"  loop = {coef[0]:.3f} us x tiles + {coef[1]:.2f} us   (tiles 1..{ntiles.max()})")
print(f"  epilogue (loop end -> exit) median {np.median(epi):.2f} us  max {epi.max():.2f}")
for lo, hi in ((1, 8), (8, 64), (64, 256), (256, 100000)):
    m = (ntiles >= lo) & (ntiles < hi)
    if m.any():
        print(f"  tiles in [{lo},{hi}): n={m.sum():5d}  loop/tile median {np.median(loop[m]/ntiles[m]):.3f} us   workgroup duration median {np.median(dur[m]):.1f} us")
mhz = (clk[:, 2] - clk[:, 1]) / ((a[:, 2] - a[:, 1]) / 100.0)
big = ntiles >= 32
print(f"  s_memtime ticks per us of s_memrealtime over the loop (workgroups of >= 32 tiles): median {np.median(mhz[big]):.0f}  min {mhz[big].min():.0f}  max {mhz[big].max():.0f}")
# idle gaps per CU cannot be seen here; total busy time vs span:
print(f"  sum of workgroup durations / 256 CUs = {dur.sum()/256:.1f} us")
order = np.argsort(rel[:, 0])
print("  last 5 workgroups to start: start, tiles, end:", [(round(rel[i,0],1), int(ntiles[i]), round(rel[i,3],1)) for i in order[-5:]])
# successor gap: the k-th workgroup to start (k >= 256) takes the slot of the (k - 256)-th to finish
st = np.sort(rel[:, 0]); en = np.sort(rel[:, 3])
if len(st) > 512:
    gap = st[256:] - en[:len(st) - 256]
    print(f"  slot hand-over (k-th start minus (k-256)-th exit): median {np.median(gap):.2f} us  p10 {np.percentile(gap,10):.2f}  p90 {np.percentile(gap,90):.2f}")
    print(f"  first 256 starts span {st[255]-st[0]:.2f} us; last exit {en[-1]:.1f} us, 256-th last exit {en[-256]:.1f} us")
