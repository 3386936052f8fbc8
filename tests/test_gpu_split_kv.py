"""GPU: split-KV decode shards.  One process plays W devices: the rows of every (b, h) are dealt
page by page to W separate paged caches, each "device" runs the product's decode attention with `return_lse=True`
on its slice, the shard results are merged by the HIP merge kernel - against the oracle's full attention and the
oracle's per-shard (out, lse).  The collective itself is covered over gloo in tests/test_cpu_distributed.py."""
import math

import pytest
import torch

from helpers import mk_paged, tol
from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu


def _shard(pt, bm, lens, PS, W, r):
    my_pt, my_lens = torch.zeros_like(pt), torch.zeros_like(lens)
    B, HKV = lens.shape
    for b in range(B):
        for h in range(HKV):
            L = int(lens[b, h])
            pages = [p for p in range(-(-L // PS)) if p % W == r]
            for j, p in enumerate(pages):
                my_pt[int(bm[b]), h, j] = pt[int(bm[b]), h, p]
            my_lens[b, h] = sum(min(PS, L - p * PS) for p in pages)
    return my_pt, my_lens


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,HQ,HKV,D,PS,maxlen,W,key_split", [
    (1, 32, 8, 128, 128, 9000, 2, None),   # one long sequence over 2 devices: in-launch merge inside every shard
    (2, 8, 2, 64, 128, 3000, 3, None),     # 3 devices, D = 64, G = 4
    (40, 32, 8, 128, 128, 700, 2, None),   # 320 (b,h) groups: one split per group (direct LSE output)
    (3, 16, 4, 128, 128, 5000, 2, 40),     # oversubscribed grid: two-kernel merge inside every shard
])
def test_split_kv_shards_merge_to_full_attention(dev, dtype, B, HQ, HKV, D, PS, maxlen, W, key_split):
    from compactor_vllm_amd.attention.cross_gpu_decode import merge_shards
    from compactor_vllm_amd.attention.sparse_decode_kernel import head_sparse_decode_attention

    g = torch.Generator().manual_seed(B + maxlen + W)
    lens = torch.randint(1, maxlen + 1, (B, HKV), generator=g, dtype=torch.int32)
    lens[0, 0] = maxlen
    lens[-1, -1] = 60  # fits one page: every device but one holds nothing of this head
    if B > 1:
        lens[1, 0] = 0  # nobody holds anything
    kc, vc, pt, bm, P = mk_paged(B, HKV, D, PS, lens, dtype, seed=3)
    q = torch.randn(B, HQ, D, generator=g).to(dtype)
    scale = 1.0 / math.sqrt(D)
    kcd, vcd, qd, bmd = kc.to(dev), vc.to(dev), q.to(dev), bm.to(dev)
    outs, lses = [], []
    for r in range(W):
        my_pt, my_lens = _shard(pt, bm, lens, PS, W, r)
        o, l = head_sparse_decode_attention(qd, kcd, vcd, my_lens.to(dev), my_pt.to(dev), bmd, HKV, PS, scale,
                                            key_split=key_split, return_lse=True)
        o_ref, l_ref = O.decode_attention_lse(q, kc, vc, my_lens, my_pt, bm, HKV, PS, scale)
        fin = torch.isfinite(l_ref)
        assert torch.equal(fin, torch.isfinite(l.cpu())), r
        assert torch.allclose(l.cpu()[fin], l_ref[fin], rtol=1e-4, atol=1e-3), (l.cpu()[fin] - l_ref[fin]).abs().max()
        assert torch.allclose(o.cpu().float(), o_ref.float(), atol=tol(dtype))
        outs.append(o)
        lses.append(l)
    merged = merge_shards(torch.stack(outs), torch.stack(lses))
    torch.cuda.synchronize()
    ref = O.decode_attention(q, kc, vc, lens, pt, bm, HKV, PS, scale).float()
    assert torch.allclose(merged.cpu().float(), ref, atol=tol(dtype)), (merged.cpu().float() - ref).abs().max()
    ref_m = O.merge_shards(torch.stack(outs).cpu(), torch.stack(lses).cpu()).float()
    assert torch.allclose(merged.cpu().float(), ref_m, atol=tol(dtype) / 4)
    if B > 1:
        G = HQ // HKV
        assert (merged[1, :G] == 0).all()
