"""Randomised soak of the 4-wave prefill kernel against the CPU oracle (not part of the test suite: ~200 random shapes,
tile-aligned and off-by-one lengths, empty / partial cached prefixes, both dtypes, G in {1,2,4,8}, page sizes 64/128/256).
python tools/dbg/soak_prefill.py [cases] [seed]"""
import math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "compactor-vllm_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from helpers import mk_paged, tol
from oracle import ref_cpu as O
from compactor_vllm_amd.attention.sparse_varlen_kernel import causal_sparse_varlen_with_cache

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
g = torch.Generator().manual_seed(seed)
dev = torch.device("cuda:0")
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
special = [1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257, 320, 511, 512, 513]
worst = 0.0
for ci in range(cases):
    dtype = [torch.float16, torch.bfloat16][ri(0, 1)]
    G = [1, 2, 4, 8][ri(0, 3)]
    HKV = [1, 2, 4][ri(0, 2)]
    HQ, D = HKV * G, 128
    PS = [64, 128, 256][ri(0, 2)]
    B = ri(1, 3)
    append = [special[ri(0, len(special) - 1)] if ri(0, 1) else ri(1, 600) for _ in range(B)]
    cache_max = [0, special[ri(0, len(special) - 1)], ri(1, 700)][ri(0, 2)]
    lens = torch.zeros(B, HKV, dtype=torch.int32)
    if cache_max:
        for b in range(B):
            for h in range(HKV):
                lens[b, h] = [0, cache_max, ri(0, cache_max), special[ri(0, len(special) - 1)] % (cache_max + 1)][ri(0, 3)]
    kc, vc, pt, bm, P = mk_paged(B, HKV, D, PS, lens, dtype, seed=ci + 100)
    cu = torch.tensor([0] + torch.tensor(append).cumsum(0).tolist(), dtype=torch.int32)
    N = int(cu[-1])
    scale_q = [1.0, 1.0, 4.0][ri(0, 2)]
    q = (torch.randn(N, HQ, D, generator=g) * scale_q).to(dtype)
    k = torch.randn(N, HKV, D, generator=g).to(dtype)
    v = torch.randn(N, HKV, D, generator=g).to(dtype)
    scale = 1.0 / math.sqrt(D)
    out = causal_sparse_varlen_with_cache(q.to(dev), k.to(dev), v.to(dev), kc.to(dev), vc.to(dev), lens.to(dev), pt.to(dev),
                                          bm.to(dev), cu.to(dev), max(append), int(lens.max()), HKV, PS, scale)
    torch.cuda.synchronize()
    ref = O.prefill_attention(q, k, v, kc, vc, lens, pt, bm, cu, HKV, PS, scale).float()
    o = out.cpu().float()
    d = float((o - ref).abs().max())
    worst = max(worst, float(((o - ref).abs() / (tol(dtype) + (2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10) * ref.abs())).max()))
    # the oracle returns the model dtype: where |o| >= 2 one ulp of the OUTPUT rounding (2^-7 |o| in bf16, 2^-10 in fp16)
    # is already above the attention tolerance, so one ulp of the reference value is allowed on top of it
    rt = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    ok = torch.isfinite(o).all() and torch.allclose(o, ref, rtol=rt, atol=tol(dtype))
    if not ok:
        print(f"FAIL case {ci}: dtype={dtype} G={G} HKV={HKV} PS={PS} append={append} lens={lens.tolist()} maxdiff={d}")
        sys.exit(1)
print(f"soak ok: {cases} cases, worst |diff| / tolerance = {worst:.3f}")
