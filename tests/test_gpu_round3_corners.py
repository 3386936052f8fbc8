"""GPU parity (through the C ABI) for the corners SURVEY 8(c) lists and round 2 left open: SnapKV at the reference's
other two autotune outcomes (pool tile 32 / 64), decode attention with RESERVED_BATCH padding rows, the C1 4096-token
dense prefill of BASELINE.json configs[0] - all against vectors produced by the reference itself - and the two sticky
status words of the library (selection look-back, in-launch decode merge)."""
import ctypes
import os

import pytest
import torch

from golden_io import list_cases, load_case
from helpers import tol
from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", list_cases("snapkvt_"))
def test_snapkv_golden_pool_tiles(dev, name):
    """`pool_tile` = the reference's autotuned BLOCK_K (snapkv.py:160-168, :253-262): vectors made with the reference
    pinned to 32 and to 64; the default (128) must NOT reproduce them."""
    from compactor_vllm_amd.compression.snapkv import query_aware_key_scores

    c = load_case(name)
    cu = c["cu_seqlens"].to(dev)
    q, k = c["q"].to(dev), c["k"].to(dev)
    out = query_aware_key_scores(q, k, cu, cu, w=c["w"], pool_tile=c["pool_tile"]).cpu()
    out128 = query_aware_key_scores(q, k, cu, cu, w=c["w"]).cpu()
    orc = O.snapkv_scores(c["q"], c["k"], c["cu_seqlens"], c["cu_seqlens"], c["w"], pool_tile=c["pool_tile"])
    ref = c["out"]
    s, differs = 0, False
    for L in c["cu_seqlens"].diff().tolist():
        a, r = out[s : s + L], ref[s : s + L]
        if L > c["w"]:
            fin = torch.isfinite(r)
            assert torch.equal(fin, torch.isfinite(a))
            assert torch.allclose(a[fin], r[fin], rtol=2e-4, atol=2e-5), (a[fin] - r[fin]).abs().max()
            assert torch.allclose(a[fin], orc[s : s + L][fin], rtol=2e-4, atol=2e-5)
            differs |= not torch.allclose(out128[s : s + L][fin], r[fin], rtol=1e-3, atol=1e-3)
        else:
            assert torch.isinf(a).all()
        s += L
    assert differs


def test_snapkv_pool_tile_is_validated(dev):
    from compactor_vllm_amd.compression.snapkv import query_aware_key_scores

    q = torch.randn(64, 4, 128, device=dev, dtype=torch.float16)
    cu = torch.tensor([0, 64], dtype=torch.int32, device=dev)
    with pytest.raises(AssertionError):
        query_aware_key_scores(q, q[:, :2].contiguous(), cu, cu, w=8, pool_tile=48)


@pytest.mark.parametrize("name", list_cases("decoderes_"))
@pytest.mark.parametrize("key_split", [None, 1, 3])
def test_decode_golden_reserved_rows(dev, name, key_split):
    """The engine's graph path pads the batch with RESERVED_BATCH entries of length 0 (model_runner.py:468-491): the live
    rows must equal the reference's, the padded rows are zeros here (uninitialised upstream, Q6) and the cache is not
    touched."""
    from compactor_vllm_amd.attention.sparse_decode_kernel import head_sparse_decode_attention

    c = load_case(name)
    q, kc, vc, lens, pt, bm = (c[n].to(dev) for n in ("q", "k_cache", "v_cache", "seq_lens_bh", "page_table",
                                                      "batch_mapping"))
    out = head_sparse_decode_attention(q, kc, vc, lens, pt, bm, c["HKV"], c["PAGE_SIZE"], c["sm_scale"],
                                       key_split=key_split).cpu()
    live = c["live_rows"].bool()
    assert torch.allclose(out[live].float(), c["out"][live].float(), rtol=1e-6, atol=tol(q.dtype))
    assert (out[~live] == 0).all()
    assert torch.equal(kc.cpu(), c["k_cache"]) and torch.equal(vc.cpu(), c["v_cache"])


def test_prefill_c1_4096_reference_rows(dev):
    """BASELINE.json configs[0] (C1): 4096-token dense causal prefill, HQ 32 / HKV 8 / D 128 / page 128, fp16, against
    512 output rows of the reference's own kernel (tests/golden/gen_fixtures.py c1) at the reference's own bar
    (atol 3e-3, tests/test_triton_attention.py:283); both kernel structures."""
    from test_oracle_golden import c1_inputs

    from compactor_vllm_amd.attention.sparse_varlen_kernel import causal_sparse_varlen_with_cache

    c = load_case("c1prefill_4096")
    q, k, v = c1_inputs(c)
    HKV, PS, D, N = c["HKV"], c["PAGE_SIZE"], c["D"], c["N"]
    kc = torch.zeros(PS, D, dtype=torch.float16, device=dev)
    args = (kc, kc.clone(), torch.zeros(1, HKV, dtype=torch.int32, device=dev),
            torch.zeros(2, HKV, 1, dtype=torch.int32, device=dev), torch.tensor([1], dtype=torch.int32, device=dev),
            torch.tensor([0, N], dtype=torch.int32, device=dev), N, 0, HKV, PS, c["sm_scale"])
    out = causal_sparse_varlen_with_cache(q.to(dev), k.to(dev), v.to(dev), *args).cpu()
    rows = out[c["tok"].long(), c["head"].long()]
    assert torch.allclose(rows.float(), c["rows"].float(), rtol=1e-6, atol=3e-3), (rows.float() - c["rows"].float()).abs().max()
    assert abs(float(out.float().abs().mean()) - c["out_abs_mean"]) < 1e-3


def test_select_lookback_timeout_raises_status_not_trap(dev):
    """Boundary contract: the library never traps.  A debug build of select.hip (tools/dbg/build_test_libs.py,
    -DCVLLM_SEL_WITHHOLD) withholds slice 0's look-back word of every column: the waiting slices must give up after
    their bounded wait, raise the sticky error word (`cvllm_select_status` -> 1, then cleared -> 0), keep every written
    index a valid token, and the process must survive; the product library on the same input reports 0."""
    from compactor_vllm_amd import _lib
    from compactor_vllm_amd.compression.common import select_retained, select_status

    path = os.path.join(ROOT, "tools", "dbg", "libcvllm_sel_withhold.so")
    if not os.path.exists(path):
        pytest.fail("tools/dbg/libcvllm_sel_withhold.so missing: run __graft_entry__.build()")
    L, H, PS = 3 * 4096 + 100, 8, 128  # >= 8192 tokens: the multi-slice per-head path, 4 slices per column
    g = torch.Generator(device=dev).manual_seed(3)
    scores = torch.randn(L, H, device=dev, generator=g)
    cu = torch.tensor([0, L], dtype=torch.int32, device=dev)
    retain = torch.tensor([L * H // 2], dtype=torch.int32, device=dev)
    bm = torch.ones(1, dtype=torch.int32, device=dev)
    l0 = torch.zeros(1, H, dtype=torch.int32, device=dev)
    # product library: healthy
    kept_ok, lens_ok = select_retained(scores, cu, L, retain, bm, l0, PS, True)
    assert select_status() == 0
    dbg = ctypes.CDLL(path)
    dbg.cvllm_select_workspace_bytes.restype = ctypes.c_size_t
    dbg.cvllm_select_workspace_bytes.argtypes = [ctypes.c_int] * 3
    dbg.cvllm_select_topk.restype = ctypes.c_int
    dbg.cvllm_select_topk.argtypes = _lib.SIGNATURES["cvllm_select_topk"][1]
    dbg.cvllm_select_status.restype = ctypes.c_int
    dbg.cvllm_select_status.argtypes = [ctypes.c_void_p]
    wsb = dbg.cvllm_select_workspace_bytes(1, H, L)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    kept = torch.full((1, H, L), -12345, dtype=torch.int32, device=dev)
    lens = torch.empty(1, H, dtype=torch.int32, device=dev)
    st = dbg.cvllm_select_topk(scores.data_ptr(), cu.data_ptr(), retain.data_ptr(), l0.data_ptr(), bm.data_ptr(),
                               kept.data_ptr(), lens.data_ptr(), 1, H, L, PS, 1, 0, ws.data_ptr(), wsb, _lib.stream())
    assert st == 0  # the launch itself succeeds
    assert dbg.cvllm_select_status(_lib.stream()) == 1  # ... and the failure is reported, not trapped on
    assert dbg.cvllm_select_status(_lib.stream()) == 0  # sticky word cleared by the check
    assert torch.equal(lens, lens_ok)  # the counts come from the histogram passes, which were complete
    written = kept[kept != -12345]
    assert written.numel() > 0 and int(written.min()) >= 0 and int(written.max()) < L
    # the product path still works in the same process afterwards (no context loss)
    kept2, lens2 = select_retained(scores, cu, L, retain, bm, l0, PS, True)
    assert torch.equal(lens2, lens_ok) and select_status() == 0
    for h in range(H):  # entries beyond a head's count are unspecified (torch.empty)
        n = int(lens_ok[0, h])
        assert torch.equal(kept2[0, h, :n], kept_ok[0, h, :n])


@torch.inference_mode()  # like the engine: graphs captured earlier in the process registered inference tensors
def test_decode_merge_modes_and_status_over_all_workspaces(dev):
    """The default split merge is the two-kernel path; `merge_status` looks at every live decode workspace of the device
    (eager stream AND graph capture stream) and stays 0."""
    from compactor_vllm_amd.attention import sparse_decode_kernel as dk
    from helpers import mk_paged

    B, HQ, HKV, D, PS = 1, 32, 8, 128, 128
    lens = torch.full((B, HKV), 3000, dtype=torch.int32)
    kc, vc, pt, bm, P = mk_paged(B, HKV, D, PS, lens, torch.bfloat16, seed=3)
    q = torch.randn(B, HQ, D).to(torch.bfloat16)
    args = (kc.to(dev), vc.to(dev), lens.to(dev), pt.to(dev), bm.to(dev), HKV, PS)
    qd = q.to(dev)
    out = dk.head_sparse_decode_attention(qd, *args)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        dk.head_sparse_decode_attention(qd, *args)  # allocates the side stream's workspace
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            out_g = dk.head_sparse_decode_attention(qd, *args)
    g.replay()
    torch.cuda.synchronize()
    ref = O.decode_attention(q, kc, vc, lens, pt, bm, HKV, PS, 1.0 / D ** 0.5)
    assert torch.allclose(out.cpu().float(), ref.float(), atol=2e-2)
    assert torch.allclose(out_g.cpu().float(), ref.float(), atol=2e-2)
    live = [k for k in dk._workspaces if k[0] == dev.index]
    assert len(live) >= 2  # current stream + capture stream
    assert dk.merge_status(dev) == 0
    for k in live:  # contract: every completed call leaves its workspace all zeros
        assert int(dk._workspaces[k].count_nonzero()) == 0
