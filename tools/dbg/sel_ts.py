"""Phase stamps of the joint slice-histogram passes (debug library with -DCVLLM_SEL_TS):  python tools/dbg/sel_ts.py <lib.so>"""
import ctypes, os, sys
import numpy as np
os.environ["CVLLM_LIB_PATH"] = os.path.abspath(sys.argv[1])
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "compactor-vllm_amd"))
import torch
from compactor_vllm_amd import _lib
from compactor_vllm_amd.compression.common import select_retained
L, HKV = 32768, 8
torch.manual_seed(0)
sc = torch.randn(L, HKV, device="cuda")
cu = torch.tensor([0, L], device="cuda", dtype=torch.int32)
retain = torch.tensor([L * HKV // 2], dtype=torch.int32, device="cuda")
zero = torch.zeros(1, HKV, dtype=torch.int32, device="cuda")
bm = torch.ones(1, dtype=torch.int32, device="cuda")
for _ in range(5):
    select_retained(sc, cu, L, retain, bm, zero, 128)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (8 * 64 * 8))()
_lib.lib().cvllm_debug_select_stamps(buf)
a = np.frombuffer(buf, dtype=np.uint64).reshape(8, 64, 8).astype(np.int64)
t0 = a[0, :32, 0].min()
print("sj_hist_kernel passes 0..2, 32 slice workgroups; us after pass 0's first entry:")
print("  stamps: entry, state read, LDS histogram done, global atomics issued, ticket known, [last arriver: radix step done]")
for ps in range(3):
    t = (a[ps, :32, :5] - t0) / 100.0
    last = a[ps, :32, 5].max()
    print("  pass %d median" % ps, np.round(np.median(t, axis=0), 2), " max", np.round(t.max(axis=0), 2), " last arriver done %.2f" % ((last - t0) / 100.0))
