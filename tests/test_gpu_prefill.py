"""GPU parity (through the C ABI): prefill attention (a1) vs golden vectors and the fp32 CPU oracle."""
import math

import pytest
import torch

from golden_io import list_cases, load_case
from helpers import mk_paged, tol
from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list_cases("prefill_"))
def test_prefill_golden(dev, name):
    from compactor_vllm_amd.attention.sparse_varlen_kernel import causal_sparse_varlen_with_cache

    c = load_case(name)
    HQ, HKV, D = c["HQ"], c["HKV"], c["D"]
    qkv = c["qkv"].to(dev)
    N = qkv.shape[0]
    k = qkv[:, HQ * D : (HQ + HKV) * D].view(N, HKV, D)  # strided views of the fused projection
    v = qkv[:, (HQ + HKV) * D :].view(N, HKV, D)
    q = c["q"].to(dev)
    lens = c["seq_lens_bh"]
    out = causal_sparse_varlen_with_cache(
        q, k, v, c["k_cache"].to(dev), c["v_cache"].to(dev), lens.to(dev), c["page_table"].to(dev),
        c["batch_mapping"].to(dev), c["cu_seqlens_q"].to(dev), int(c["cu_seqlens_q"].diff().max()), int(lens.max()),
        HKV, c["PAGE_SIZE"], c["sm_scale"])
    torch.cuda.synchronize()
    ref = c["out"].float()
    d = (out.cpu().float() - ref).abs().max()
    assert torch.allclose(out.cpu().float(), ref, rtol=1e-6, atol=tol(q.dtype)), d


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,HQ,HKV,D,PS,cache_max,append", [
    (1, 32, 8, 128, 128, 0, [777]),              # dense causal, ragged tail tile
    (2, 32, 8, 128, 256, 300, [1, 513]),         # append_len == 1 (reference fast path) + multi-tile
    (3, 8, 8, 128, 128, 140, [64, 65, 2]),       # G = 1
    (2, 16, 2, 64, 128, 200, [130, 70]),         # G = 8, D = 64
    (2, 16, 2, 128, 128, 200, [130, 70]),        # G = 8, D = 128 (4-wave kernel: a query block = one head x 32 tokens)
    (1, 4, 4, 128, 256, 513, [300]),             # G = 1, D = 128, cached prefix longer than the appended block
    (2, 8, 4, 128, 128, 260, [256, 300]),        # G = 2
    (2, 32, 8, 128, 32, 150, [70, 200]),         # PS = 32: a 64-key tile spans two pages (per-row page lookup path)
    (2, 16, 4, 128, 96, 300, [130, 257]),        # PS = 96: tiles straddle page boundaries at varying offsets
    (1, 32, 8, 128, 64, 700, [300]),             # PS = 64: one tile per page (scalar page-walk path, many pages)
])
def test_prefill_oracle_shapes(dev, dtype, B, HQ, HKV, D, PS, cache_max, append):
    """Per-head-varying prefix lengths, shuffled pages, batch_mapping != arange, strided v (untested
    upstream)."""
    from compactor_vllm_amd.attention.sparse_varlen_kernel import causal_sparse_varlen_with_cache

    g = torch.Generator().manual_seed(B * 7 + HQ + cache_max)
    lens = torch.randint(0, cache_max + 1, (B, HKV), generator=g, dtype=torch.int32)
    if cache_max:
        lens[0, 0] = 0
        lens[-1, -1] = cache_max
    kc, vc, pt, bm, P = mk_paged(B, HKV, D, PS, lens, dtype, seed=13)
    cu = torch.tensor([0] + torch.tensor(append).cumsum(0).tolist(), dtype=torch.int32)
    N = int(cu[-1])
    q = torch.randn(N, HQ, D, generator=g).to(dtype)
    qkv = torch.randn(N, (HQ + 2 * HKV) * D, generator=g).to(dtype)
    k = qkv[:, HQ * D : (HQ + HKV) * D].view(N, HKV, D)
    v = qkv[:, (HQ + HKV) * D :].view(N, HKV, D)
    scale = 1.0 / math.sqrt(D)
    qkv_d = qkv.to(dev)
    out = causal_sparse_varlen_with_cache(
        q.to(dev), qkv_d[:, HQ * D : (HQ + HKV) * D].view(N, HKV, D), qkv_d[:, (HQ + HKV) * D :].view(N, HKV, D),
        kc.to(dev), vc.to(dev), lens.to(dev), pt.to(dev), bm.to(dev), cu.to(dev), max(append), int(lens.max()), HKV, PS,
        scale)
    torch.cuda.synchronize()
    ref = O.prefill_attention(q, k, v, kc, vc, lens, pt, bm, cu, HKV, PS, scale)
    d = (out.cpu().float() - ref.float()).abs().max()
    assert torch.allclose(out.cpu().float(), ref.float(), rtol=1e-6, atol=tol(dtype)), d


def test_prefill_c1_shape_4k(dev):
    """BASELINE.json configs[0]: HQ=32 HKV=8 D=128 page=128, 4096 tokens, bs=1, fp16, dense causal.
    Checked on sampled rows against the oracle (the full 4K oracle is minutes of CPU) and through the
    row-sum property: attention of V == const gives that const."""
    from compactor_vllm_amd.attention.sparse_varlen_kernel import causal_sparse_varlen_with_cache

    dtype, B, HQ, HKV, D, PS, N = torch.float16, 1, 32, 8, 128, 128, 4096
    g = torch.Generator().manual_seed(1234)
    lens = torch.zeros(B, HKV, dtype=torch.int32)
    kc, vc, pt, bm, P = mk_paged(B, HKV, D, PS, lens, dtype, seed=1)
    cu = torch.tensor([0, N], dtype=torch.int32)
    q = torch.randn(N, HQ, D, generator=g).to(dtype)
    k = torch.randn(N, HKV, D, generator=g).to(dtype)
    v = torch.randn(N, HKV, D, generator=g).to(dtype)
    args = (kc.to(dev), vc.to(dev), lens.to(dev), pt.to(dev), bm.to(dev), cu.to(dev), N, 0, HKV, PS)
    out = causal_sparse_varlen_with_cache(q.to(dev), k.to(dev), v.to(dev), *args).cpu().float()
    rows = [0, 1, 63, 64, 255, 256, 1000, 2047, 4095]
    G = HQ // HKV
    for t in rows:
        for hq in (0, 5, 31):
            kk = k[: t + 1, hq // G].float()
            p = torch.softmax(q[t, hq].float() @ kk.T / math.sqrt(D), -1)
            ref = p @ v[: t + 1, hq // G].float()
            assert torch.allclose(out[t, hq], ref, atol=3e-3), (t, hq)
    ones = torch.ones_like(v).to(dev)
    out1 = causal_sparse_varlen_with_cache(q.to(dev), k.to(dev), ones, *args).cpu().float()
    assert torch.allclose(out1, torch.ones_like(out1), atol=2e-3)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_prefill_deterministic_and_large_logits(dev, dtype):
    """Two launches on the same inputs are bit-identical (the running max is carried across tiles in registers:
    a stale read would only change roundings), also while another stream keeps the GPU busy; logits that jump
    by orders of magnitude between tiles (scaled q rows) stay finite and match the oracle."""
    from compactor_vllm_amd.attention.sparse_varlen_kernel import causal_sparse_varlen_with_cache

    B, HQ, HKV, D, PS = 2, 8, 2, 128, 128
    g = torch.Generator().manual_seed(99)
    lens = torch.tensor([[70, 0], [200, 131]], dtype=torch.int32)
    kc, vc, pt, bm, P = mk_paged(B, HKV, D, PS, lens, dtype, seed=5)
    append = [513, 129]
    cu = torch.tensor([0, 513, 642], dtype=torch.int32)
    N = 642
    q = torch.randn(N, HQ, D, generator=g)
    q[::7] *= 6.0  # every 7th query: logits ~ 6x larger -> the row max moves a lot from tile to tile
    k = torch.randn(N, HKV, D, generator=g)
    k[300:310] *= 4.0
    v = torch.randn(N, HKV, D, generator=g)
    q, k, v = q.to(dtype), k.to(dtype), v.to(dtype)
    scale = 1.0 / math.sqrt(D)
    args = (kc.to(dev), vc.to(dev), lens.to(dev), pt.to(dev), bm.to(dev), cu.to(dev), max(append), int(lens.max()),
            HKV, PS, scale)
    qd, kd, vd = q.to(dev), k.to(dev), v.to(dev)
    out1 = causal_sparse_varlen_with_cache(qd, kd, vd, *args)
    side = torch.cuda.Stream()
    junk = torch.randn(2048, 2048, device=dev)
    with torch.cuda.stream(side):
        for _ in range(8):
            junk = junk @ junk * 1e-3
    out2 = causal_sparse_varlen_with_cache(qd, kd, vd, *args)
    torch.cuda.synchronize()
    assert torch.equal(out1, out2)
    assert torch.isfinite(out1.float()).all()
    ref = O.prefill_attention(q, k, v, kc, vc, lens, pt, bm, cu, HKV, PS, scale)
    d = (out1.cpu().float() - ref.float()).abs().max()
    assert torch.allclose(out1.cpu().float(), ref.float(), rtol=1e-6, atol=tol(dtype)), d


@pytest.mark.parametrize("split", [1024, 1000])
def test_prefill_chunked_equals_one_shot(dev, split):
    """SURVEY 8f-3 property: prefill in two chunks (chunk 1 written to the paged cache with prefill_store_all_kv,
    chunk 2 attending to [cached prefix || itself]) equals the one-shot causal prefill within the attention tolerance
    (the softmax is the same sum in a different tile order)."""
    from compactor_vllm_amd.attention.sparse_varlen_kernel import causal_sparse_varlen_with_cache
    from compactor_vllm_amd.kv_cache.store_kv_cache import prefill_store_all_kv

    dtype, HQ, HKV, D, PS, N = torch.bfloat16, 32, 8, 128, 128, 2048
    g = torch.Generator().manual_seed(7)
    q = torch.randn(N, HQ, D, generator=g).to(dtype).to(dev)
    k = torch.randn(N, HKV, D, generator=g).to(dtype).to(dev)
    v = torch.randn(N, HKV, D, generator=g).to(dtype).to(dev)
    P = N // PS
    pt = torch.randperm(2 * HKV * P, generator=g).view(2, HKV, P).to(torch.int32).to(dev)
    kc = torch.zeros(2 * HKV * P * PS, D, dtype=dtype, device=dev)
    vc = torch.zeros_like(kc)
    bm = torch.ones(1, dtype=torch.int32, device=dev)
    zero = torch.zeros(1, HKV, dtype=torch.int32, device=dev)

    cu = torch.tensor([0, N], dtype=torch.int32, device=dev)
    one = causal_sparse_varlen_with_cache(q, k, v, kc, vc, zero, pt, bm, cu, N, 0, HKV, PS)

    cu1 = torch.tensor([0, split], dtype=torch.int32, device=dev)
    o1 = causal_sparse_varlen_with_cache(q[:split], k[:split], v[:split], kc, vc, zero, pt, bm, cu1, split, 0, HKV, PS)
    lens = zero.clone()
    prefill_store_all_kv(new_keys=k[:split], new_values=v[:split], cu_seqlens_k=cu1, max_seqlen_k=split, k_cache=kc,
                         v_cache=vc, page_table=pt, bh_lens=lens, batch_mapping=bm, PAGE_SIZE=PS)
    assert int(lens.min()) == split and int(lens.max()) == split
    rest = N - split
    cu2 = torch.tensor([0, rest], dtype=torch.int32, device=dev)
    o2 = causal_sparse_varlen_with_cache(q[split:], k[split:], v[split:], kc, vc, lens, pt, bm, cu2, rest, split, HKV, PS)
    torch.cuda.synchronize()
    two = torch.cat([o1, o2])
    d = (one.float() - two.float()).abs().max()
    assert torch.allclose(one.float(), two.float(), rtol=1e-6, atol=tol(dtype)), d
    # oracle leg: the CPU restatement's ONE-SHOT dense causal attention is what both must equal
    dummy = torch.zeros(PS, D, dtype=dtype)
    ref = O.prefill_attention(q.cpu(), k.cpu(), v.cpu(), dummy, dummy, torch.zeros(1, HKV, dtype=torch.int32),
                              torch.zeros(2, HKV, 1, dtype=torch.int32), torch.ones(1, dtype=torch.int32),
                              torch.tensor([0, N], dtype=torch.int32), HKV, PS, 1.0 / math.sqrt(D)).float()
    assert torch.allclose(two.cpu().float(), ref, rtol=1e-6, atol=tol(dtype)), (two.cpu().float() - ref).abs().max()
    assert torch.allclose(one.cpu().float(), ref, rtol=1e-6, atol=tol(dtype))


def _both_structures(fn):
    """Run fn() with the default dispatch (4-wave kernel where it applies) and with CVLLM_PREFILL=8wave."""
    import os

    out4 = fn()
    os.environ["CVLLM_PREFILL"] = "8wave"
    try:
        out8 = fn()
    finally:
        del os.environ["CVLLM_PREFILL"]
    torch.cuda.synchronize()
    return out4, out8


@pytest.mark.parametrize("jump", [30.0, 300.0])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_prefill_both_structures_and_deferred_rescale(dev, dtype, jump):
    """The 4-wave kernel computes probabilities against a per-row reference point that follows the row max only when a
    check of the packed probabilities finds one >= 2 (a logit bias + 1 above the row max of the last update; 32-key
    units), then replays the unit; the 8-wave kernel defers its rescale until the exact row max has grown by more than
    2^8, per 64-key tile.  A bounded random input never takes those branches after the first tiles, so this input is
    built to take BOTH sides many times (MI355X guide rule 26): logits along one direction that climb in small steps
    (a few units of 2^x per 32 keys: no update), jump by far more than the threshold (update + replay), fall, and climb
    again - over a cached prefix and an appended block, for query rows of different scale.  jump = 30 overflows the fp16
    probabilities of the unit that meets it (2^28), jump = 300 overflows fp32 itself (2^298): neither may reach O or l.
    Both kernels must meet the attention tolerance against the fp32 oracle; so must their difference."""
    from compactor_vllm_amd.attention.sparse_varlen_kernel import causal_sparse_varlen_with_cache

    B, HQ, HKV, D, PS = 1, 8, 2, 128, 128
    g = torch.Generator().manual_seed(321)
    Lc, N = 700, 1500
    lens = torch.tensor([[Lc, Lc - 37]], dtype=torch.int32)
    kc, vc, pt, bm, P = mk_paged(B, HKV, D, PS, lens, dtype, seed=11)
    u = torch.randn(D, generator=g)
    u = u / u.norm()
    scale = 1.0 / math.sqrt(D)

    def staircase(n, start):
        # target logit (exp2 domain) of key j along u for a query q = qs * u: steps of +2.5 per 32 keys, a +jump step
        # every 400 keys (key 400 k + 16 of a unit: in the middle of it), a drop of 4/3 of it every 700
        j = torch.arange(n) + start
        return 2.5 * (j // 32) + jump * ((j + 16) // 400) - (jump * 4 / 3) * (j // 700)

    qs = 8.0
    c = scale * 1.4426950408889634
    def keys(n, start):
        a = staircase(n, start) / (qs * c)  # so that (qs u) . (a u) * scale * log2e = staircase
        return a[:, None] * u[None, :] + 0.05 * torch.randn(n, D, generator=g)

    # cached prefix rows of both heads follow the staircase too
    for h in range(HKV):
        L = int(lens[0, h])
        rows = O.cache_rows(pt[int(bm[0]), h], L, PS)
        kc[rows] = keys(L, 0).to(dtype)
    k = torch.stack([keys(N, Lc), keys(N, Lc + 13)], dim=1).to(dtype)
    v = torch.randn(N, HKV, D, generator=g).to(dtype)
    q = (qs * u)[None, None, :] * (0.5 + torch.rand(N, HQ, 1, generator=g)) + 0.1 * torch.randn(N, HQ, D, generator=g)
    q = q.to(dtype)
    cu = torch.tensor([0, N], dtype=torch.int32)
    args = (kc.to(dev), vc.to(dev), lens.to(dev), pt.to(dev), bm.to(dev), cu.to(dev), N, int(lens.max()), HKV, PS, scale)
    qd, kd, vd = q.to(dev), k.to(dev), v.to(dev)
    out4, out8 = _both_structures(lambda: causal_sparse_varlen_with_cache(qd, kd, vd, *args))
    ref = O.prefill_attention(q, k, v, kc, vc, lens, pt, bm, cu, HKV, PS, scale).float()
    for name, o in (("4-wave", out4), ("8-wave", out8)):
        o = o.cpu().float()
        assert torch.isfinite(o).all(), name
        assert torch.allclose(o, ref, rtol=1e-6, atol=tol(dtype)), (name, (o - ref).abs().max())
    assert torch.allclose(out4.float(), out8.float(), rtol=1e-6, atol=tol(dtype))


def test_prefill_both_structures_random(dev):
    """Every dispatch condition of the 4-wave kernel (D = 128, page size % 64 == 0) on a ragged batch with per-head
    prefix lengths: both kernels against the oracle."""
    from compactor_vllm_amd.attention.sparse_varlen_kernel import causal_sparse_varlen_with_cache

    dtype, B, HQ, HKV, D, PS = torch.bfloat16, 3, 32, 8, 128, 128
    g = torch.Generator().manual_seed(17)
    lens = torch.randint(0, 400, (B, HKV), generator=g, dtype=torch.int32)
    lens[1, 3] = 0
    lens[2, 0] = 64   # a full last cached tile
    lens[0, 5] = 33   # one key into the second unit of a tile
    kc, vc, pt, bm, P = mk_paged(B, HKV, D, PS, lens, dtype, seed=3)
    append = [1, 190, 321]
    cu = torch.tensor([0] + torch.tensor(append).cumsum(0).tolist(), dtype=torch.int32)
    N = int(cu[-1])
    q = torch.randn(N, HQ, D, generator=g).to(dtype)
    k = torch.randn(N, HKV, D, generator=g).to(dtype)
    v = torch.randn(N, HKV, D, generator=g).to(dtype)
    scale = 1.0 / math.sqrt(D)
    args = (kc.to(dev), vc.to(dev), lens.to(dev), pt.to(dev), bm.to(dev), cu.to(dev), max(append), int(lens.max()),
            HKV, PS, scale)
    qd, kd, vd = q.to(dev), k.to(dev), v.to(dev)
    out4, out8 = _both_structures(lambda: causal_sparse_varlen_with_cache(qd, kd, vd, *args))
    ref = O.prefill_attention(q, k, v, kc, vc, lens, pt, bm, cu, HKV, PS, scale).float()
    for name, o in (("4-wave", out4), ("8-wave", out8)):
        o = o.cpu().float()
        assert torch.allclose(o, ref, rtol=1e-6, atol=tol(dtype)), (name, (o - ref).abs().max())
