"""GPU parity (through the C ABI) for the decode kernel's split handling: the in-launch merge (all splits of a
(batch, kv-head) resident, partials exchanged through the zeroed workspace), the two-kernel merge it falls back to on
oversubscribed grids, page tables wider than the 512-page register window, and the workspace contract."""
import math

import pytest
import torch

from helpers import mk_paged, tol
from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["two-kernel", "in-launch"])
def merge_mode(request, dev):
    """Both split merges for grids that fit the chip: the default two-kernel path and the opt-in in-launch form
    (CVLLM_DECODE_MERGE=in-launch / cvllm_decode_set_merge_mode)."""
    from compactor_vllm_amd.attention import sparse_decode_kernel as dk

    prev = dk.set_merge_mode(request.param)
    yield request.param
    dk.set_merge_mode(prev)


def _ws_is_zero(dev):
    from compactor_vllm_amd.attention import sparse_decode_kernel as dk

    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    buf = dk._workspaces.get(key)
    return buf is None or int(buf.count_nonzero()) == 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,HQ,HKV,D,PS,maxlen", [
    (1, 32, 8, 128, 128, 9000),    # the metric's launch shape: 8 groups x 32 splits, one workgroup per CU
    (3, 32, 8, 128, 128, 4000),    # 24 groups -> 10 splits (not a power of two): slices of 52 outputs
    (5, 16, 2, 128, 128, 7000),    # 10 groups -> 25 splits, G = 8: a merger's slice spans < 1 head
    (2, 4, 4, 64, 128, 33000),     # G = 1, D = 64: 64 outputs per group over 32 splits (2 per merging workgroup)
    (1, 8, 1, 128, 256, 70000),    # one group, 256 splits would exceed the cap: 128 splits, G = 8
    (7, 14, 7, 64, 32, 1500),      # odd everything, page size 32
])
def test_decode_merge_paths_agree_with_oracle(dev, merge_mode, dtype, B, HQ, HKV, D, PS, maxlen):
    """The same call with the plan that fits the chip (merged by `merge_mode`) and with a key_split hint that
    oversubscribes the chip (always the two-kernel merge) against the fp32 oracle; zero-length heads and a sequence with every
    head empty included.  Afterwards the workspace is all zeros again and the merge status is clean."""
    from compactor_vllm_amd.attention import sparse_decode_kernel as dk

    g = torch.Generator().manual_seed(B * 7919 + D + maxlen)
    lens = torch.randint(1, maxlen + 1, (B, HKV), generator=g, dtype=torch.int32)
    lens[-1, -1] = maxlen
    lens[0, 0] = 0
    if B > 1:
        lens[1, :] = 0  # a whole sequence with nothing cached: every split of its groups is empty
    kc, vc, pt, bm, P = mk_paged(B, HKV, D, PS, lens, dtype, seed=3)
    q = torch.randn(B, HQ, D, generator=g).to(dtype)
    scale = 1.0 / math.sqrt(D)
    args = (q.to(dev), kc.to(dev), vc.to(dev), lens.to(dev), pt.to(dev), bm.to(dev), HKV, PS, scale)
    ref = O.decode_attention(q, kc, vc, lens, pt, bm, HKV, PS, scale).float()
    n_in = dk.plan_internal_splits(B * HKV, P * PS, None)
    out_in = dk.head_sparse_decode_attention(*args)
    big = max(2, -(-2 * dk._cus(dev.index) // (B * HKV)))  # grid > CUs -> second-kernel merge
    out_two = dk.head_sparse_decode_attention(*args, key_split=big)
    torch.cuda.synchronize()
    for o in (out_in, out_two):
        d = (o.cpu().float() - ref).abs().max()
        assert torch.allclose(o.cpu().float(), ref, rtol=1e-6, atol=tol(dtype)), (float(d), n_in, big)
    G = HQ // HKV
    assert (out_in[0, :G] == 0).all() and (out_two[0, :G] == 0).all()
    if B > 1:
        assert (out_in[1] == 0).all() and (out_two[1] == 0).all()
    assert dk.merge_status(dev) == 0
    assert _ws_is_zero(dev)


@torch.inference_mode()  # graph capture after the engine's (inference-mode) graphs in the same process
def test_decode_in_launch_merge_is_deterministic_under_replay(dev, merge_mode):
    """Back-to-back launches on one workspace (what a decode step's 32 layers and a HIP-graph replay do): every launch
    returns the same bits, eagerly and from a captured graph."""
    from compactor_vllm_amd.attention import sparse_decode_kernel as dk

    B, HQ, HKV, D, PS = 1, 32, 8, 128, 128
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(5)
    lens = torch.randint(3000, 6000, (B, HKV), generator=g, dtype=torch.int32)
    kc, vc, pt, bm, P = mk_paged(B, HKV, D, PS, lens, dtype, seed=9)
    q = torch.randn(B, HQ, D, generator=g).to(dtype)
    args = (q.to(dev), kc.to(dev), vc.to(dev), lens.to(dev), pt.to(dev), bm.to(dev), HKV, PS)
    first = dk.head_sparse_decode_attention(*args)
    outs = [dk.head_sparse_decode_attention(*args) for _ in range(40)]
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o, first)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        dk.head_sparse_decode_attention(*args)  # warm-up on the capture stream: allocates its workspace
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            outs_g = [dk.head_sparse_decode_attention(*args) for _ in range(8)]
        for _ in range(5):
            graph.replay()
    torch.cuda.synchronize()
    for o in outs_g:
        assert torch.equal(o, first)
    ref = O.decode_attention(q, kc, vc, lens, pt, bm, HKV, PS, 1.0 / math.sqrt(D)).float()
    assert torch.allclose(first.cpu().float(), ref, rtol=1e-6, atol=tol(dtype))


def _wide_cache(B, HKV, D, PS, NLP, lens, dtype, dev, seed):
    """Page table of NLP logical pages per (row, head) of which only the pages a head uses are backed by cache
    memory (a 128 K-context table with mostly short sequences), pages shuffled, batch_mapping != arange."""
    g = torch.Generator().manual_seed(seed)
    used = [[-(-int(lens[b, h]) // PS) for h in range(HKV)] for b in range(B)]
    n_pages = sum(sum(r) for r in used) + 2
    perm = torch.randperm(n_pages, generator=g).tolist()
    bm = (torch.randperm(B + 1, generator=g)[:B] + 1).to(torch.int32)
    pt = torch.full((B + 2, HKV, NLP), n_pages - 1, dtype=torch.int32)  # unused entries point at a spare page
    it = iter(perm)
    for b in range(B):
        for h in range(HKV):
            for p in range(used[b][h]):
                pt[int(bm[b]), h, p] = next(it)
    kc = torch.randn(n_pages * PS, D, device=dev, generator=torch.Generator(device=dev).manual_seed(seed)).to(dtype)
    vc = torch.randn(n_pages * PS, D, device=dev, generator=torch.Generator(device=dev).manual_seed(seed + 1)).to(dtype)
    return kc, vc, pt.to(dev), bm.to(dev)


def _torch_decode(q, kc, vc, lens, pt, bm, HKV, PS, b, h):
    G = q.shape[1] // HKV
    n = int(lens[b, h])
    i = torch.arange(n, device=q.device)
    rows = pt[int(bm[b]), h].long()[i // PS] * PS + i % PS
    K, V = kc[rows].float(), vc[rows].float()
    p = torch.softmax(q[b, h * G : (h + 1) * G].float() @ K.T / math.sqrt(q.shape[-1]), -1)
    return p @ V


@pytest.mark.parametrize("B,long_len", [(16, 131072), (32, 70000), (2, 131072)])
def test_decode_page_table_wider_than_register_window(dev, B, long_len):
    """A 128 K-context page table (1024 logical pages at page size 128) with moderate and large batches: short and
    long actual lengths.  B = 16 x 8 heads gives 2 splits of up to 512 pages + 1 (the case that used to be
    rejected); B = 32 gives ONE split of 547 pages, i.e. the register window is reloaded inside the loop."""
    from compactor_vllm_amd.attention import sparse_decode_kernel as dk

    HQ, HKV, D, PS, NLP = 32, 8, 128, 128, 1024
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(B)
    lens = torch.randint(1, 3000, (B, HKV), generator=g, dtype=torch.int32)
    lens[0, 0] = long_len
    lens[B - 1, 3] = long_len - 77
    lens[0, 1] = 0
    kc, vc, pt, bm = _wide_cache(B, HKV, D, PS, NLP, lens, dtype, dev, seed=B + 1)
    q = torch.randn(B, HQ, D, generator=g).to(dtype).to(dev)
    lens_d = lens.to(dev)
    out = dk.head_sparse_decode_attention(q, kc, vc, lens_d, pt, bm, HKV, PS)
    torch.cuda.synchronize()
    G = HQ // HKV
    for b, h in [(0, 0), (B - 1, 3), (0, 2), (B // 2, 5), (B - 1, 7)]:
        ref = _torch_decode(q, kc, vc, lens, pt, bm, HKV, PS, b, h)
        assert torch.allclose(out[b, h * G : (h + 1) * G].float(), ref, atol=2e-2), (b, h)
    assert (out[0, G : 2 * G] == 0).all()
    # the short rows against the CPU oracle, all of them
    short = [(b, h) for b in range(B) for h in range(HKV) if 0 < int(lens[b, h]) < 3000][:40]
    for b, h in short:
        ref = _torch_decode(q, kc, vc, lens, pt, bm, HKV, PS, b, h)
        assert torch.allclose(out[b, h * G : (h + 1) * G].float(), ref, atol=2e-2), (b, h)
    assert dk.merge_status(dev) == 0


def test_fused_append_with_in_launch_merge_matches_store_then_attend(dev, merge_mode):
    """cvllm_decode_append_attn (new row substituted from registers, written by the owning split, length published by
    split 0 after its mailbox filled) == decode_store_kv followed by the attention call, bit for bit, over several
    consecutive steps that cross a page boundary."""
    from compactor_vllm_amd.attention import sparse_decode_kernel as dk
    from compactor_vllm_amd.kv_cache.store_kv_cache import decode_store_kv

    B, HQ, HKV, D, PS = 2, 32, 8, 128, 128
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(21)
    lens = torch.randint(2000, 5000, (B, HKV), generator=g, dtype=torch.int32)
    lens[0, 0] = 2 * PS - 2  # crosses into a new page at the third step
    lens[1, 7] = 0
    cap = lens + 8
    kc, vc, pt, bm, P = mk_paged(B, HKV, D, PS, cap, dtype, seed=13)
    Bmax = pt.shape[0]
    full = torch.zeros(Bmax, HKV, dtype=torch.int32)
    full[bm.long()] = lens
    a = dict(kc=kc.to(dev), vc=vc.to(dev), lens=full.to(dev))
    b_ = dict(kc=kc.to(dev).clone(), vc=vc.to(dev).clone(), lens=full.to(dev).clone())
    ptd, bmd = pt.to(dev), bm.to(dev)
    for step in range(5):
        q = torch.randn(B, HQ, D, generator=g).to(dtype).to(dev)
        k1 = torch.randn(B, HKV, D, generator=g).to(dtype).to(dev)
        v1 = torch.randn(B, HKV, D, generator=g).to(dtype).to(dev)
        o_f = dk.fused_decode_step(q, k1, v1, a["kc"], a["vc"], a["lens"], ptd, bmd, HKV, PS, reserved_batch=0)
        lb = b_["lens"].index_select(0, bmd.long()).contiguous()
        decode_store_kv(key=k1, value=v1, batch_mapping=bmd, bh_lens=lb, page_table=ptd, k_cache=b_["kc"],
                        v_cache=b_["vc"], PAGE_SIZE=PS, TRITON_RESERVED_BATCH=0)
        o_s = dk.head_sparse_decode_attention(q, b_["kc"], b_["vc"], lb, ptd, bmd, HKV, PS)
        b_["lens"].index_copy_(0, bmd.long(), lb)
        torch.cuda.synchronize()
        assert torch.equal(a["lens"], b_["lens"]), step
        assert torch.equal(o_f, o_s), (step, (o_f.float() - o_s.float()).abs().max())
    assert torch.equal(a["kc"], b_["kc"]) and torch.equal(a["vc"], b_["vc"])
    assert dk.merge_status(dev) == 0 and _ws_is_zero(dev)
