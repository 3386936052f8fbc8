"""GPU parity (through the C ABI) of the fused producer step: qkv split + optional per-head q/k
RMSNorm + RoPE (+ cache write without compression) against the reference's own outputs (tests/golden/producer_*) and
the oracle restatement."""
import math

import pytest
import torch

from golden_io import list_cases, load_case
from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu


def _ulp_close(a, b, dtype):
    """equal up to about two units in the last place of the 16-bit format (a rotated value combines two normed inputs,
    each of which may sit one unit off)"""
    ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    return torch.allclose(a.float(), b.float(), rtol=2 * ulp, atol=2 * ulp)


@pytest.mark.parametrize("name", list_cases("producer_"))
def test_producer_golden(dev, name):
    from compactor_vllm_amd.layers.rotary_embedding import fused_qkv_rope

    c = load_case(name)
    HQ, HKV, D = c["HQ"], c["HKV"], c["D"]
    qkv = c["qkv"].to(dev)
    qw, kw = c.get("q_norm_w"), c.get("k_norm_w")
    q, k, v, k_pre = fused_qkv_rope(qkv, c["positions"].to(dev), c["cos_sin"].to(dev), HQ, HKV, D,
                                    None if qw is None else qw.to(dev), None if kw is None else kw.to(dev), c["eps"],
                                    want_prerope_k=True)
    torch.cuda.synchronize()
    assert q.shape == c["q_rot"].shape and k.shape == c["k_rot"].shape and q.dtype == qkv.dtype
    assert torch.equal(v.cpu().reshape(v.shape[0], -1), c["qkv"][:, (HQ + HKV) * D :])
    if not c["has_norm"]:  # element-wise fp32 arithmetic with the reference's roundings: bit for bit
        assert torch.equal(q.cpu(), c["q_rot"]) and torch.equal(k.cpu(), c["k_rot"])
        assert torch.equal(k_pre.cpu(), c["k_pre"])  # a view of the projection
    else:  # mean(x^2) is summed in a different order than torch's reduction: one last-place unit at most
        for mine, ref in ((q, c["q_rot"]), (k, c["k_rot"]), (k_pre, c["k_pre"])):
            assert _ulp_close(mine.cpu(), ref, qkv.dtype), (mine.cpu().float() - ref.float()).abs().max()
            assert float((mine.cpu() == ref).float().mean()) > 0.98


def test_rotary_embedding_module_on_strided_views(dev):
    """`RotaryEmbedding.forward(positions, q, k)` with q / k as strided views of the fused projection (what
    models/llama3.py:96-106 passes), llama3 frequency scaling: equals the oracle bit for bit."""
    from compactor_vllm_amd.layers.rotary_embedding import RotaryEmbedding

    N, HQ, HKV, D, max_pos = 777, 32, 8, 128, 2048
    scaling = ("llama3", 8.0, 1.0, 4.0, 8192)
    g = torch.Generator().manual_seed(3)
    qkv = torch.randn(N, (HQ + 2 * HKV) * D, generator=g).to(torch.bfloat16)
    pos = torch.randint(0, max_pos, (N,), generator=g)
    rope = RotaryEmbedding(D, D, max_pos, 500000.0, scaling).to(dev)
    cs = O.rope_cos_sin_cache(D, max_pos, 500000.0, scaling)
    assert torch.equal(rope.cos_sin_cache.cpu().view(max_pos, D), cs)
    qd = qkv.to(dev)
    q = qd[:, : HQ * D].view(N, HQ, D)
    k = qd[:, HQ * D : (HQ + HKV) * D].view(N, HKV, D)
    qr, kr = rope(pos.to(dev), q, k)
    ref = O.qkv_producer(qkv, pos, cs, HQ, HKV, D)
    assert torch.equal(qr.cpu(), ref[0]) and torch.equal(kr.cpu(), ref[1])


@pytest.mark.parametrize("norm", [False, True])
def test_producer_full_size_and_cache_write(dev, norm):
    """C3 / C4 layer shape (32 768 tokens, HQ 32, HKV 8, D 128): against the oracle's ops evaluated on the GPU, and the
    cache-write variant (no compression) against prefill_store_all_kv fed with the same rotated keys."""
    from compactor_vllm_amd.kv_cache.store_kv_cache import prefill_store_all_kv
    from compactor_vllm_amd.layers.rotary_embedding import CacheWrite, fused_qkv_rope

    N, HQ, HKV, D, PS, max_pos = 32768, 32, 8, 128, 128, 40960
    dtype = torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(8)
    qkv = torch.randn(N, (HQ + 2 * HKV) * D, device=dev, generator=g).to(dtype)
    lens = [20000, 1, 12767]
    pos = torch.cat([torch.arange(L, device=dev) for L in lens])
    cs = O.rope_cos_sin_cache(D, max_pos, 1000000.0).to(dev)
    qw = (1 + 0.2 * torch.randn(D, device=dev, generator=g)).to(dtype) if norm else None
    kw = (1 + 0.2 * torch.randn(D, device=dev, generator=g)).to(dtype) if norm else None
    B = len(lens)
    P = -(-max(lens) // PS)
    n_pages = (B + 1) * HKV * P
    pt = torch.randperm(n_pages, device=dev).view(B + 1, HKV, P).to(torch.int32)
    bm = torch.tensor([2, 1, 3], dtype=torch.int32, device=dev)
    cu = torch.tensor([0, 20000, 20001, N], dtype=torch.int32, device=dev)
    kc, vc = torch.zeros(n_pages * PS, D, dtype=dtype, device=dev), torch.zeros(n_pages * PS, D, dtype=dtype, device=dev)
    bh = torch.zeros(B, HKV, dtype=torch.int32, device=dev)
    q, k, v, _ = fused_qkv_rope(qkv, pos, cs, HQ, HKV, D, qw, kw, 1e-6,
                                cache_write=CacheWrite(kc, vc, cu, bm, bh, pt, PS))
    torch.cuda.synchronize()
    ref = O.qkv_producer(qkv, pos, cs, HQ, HKV, D, qw, kw, 1e-6)  # torch ops on the device: not this package's kernels
    if norm:
        assert _ulp_close(q, ref[0], dtype) and _ulp_close(k, ref[1], dtype)
    else:
        assert torch.equal(q, ref[0]) and torch.equal(k, ref[1])
    # cache: same bytes as the stand-alone store of the rotated keys
    kc2, vc2 = torch.zeros_like(kc), torch.zeros_like(vc)
    bh2 = torch.zeros_like(bh)
    prefill_store_all_kv(new_keys=k, new_values=v, cu_seqlens_k=cu, max_seqlen_k=max(lens), k_cache=kc2, v_cache=vc2,
                         page_table=pt, bh_lens=bh2, batch_mapping=bm, PAGE_SIZE=PS)
    torch.cuda.synchronize()
    assert torch.equal(bh, bh2) and torch.equal(bh.cpu(), torch.tensor(lens, dtype=torch.int32)[:, None].expand(-1, HKV))
    assert torch.equal(kc, kc2) and torch.equal(vc, vc2)
