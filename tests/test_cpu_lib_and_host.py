"""CPU suite: the C-ABI library loads and exports every symbol include/cvllm.h declares (no compute calls
without a GPU), host-side logic (chunk splitting, protected ranges, split planning, page allocator), and the
product path's refusal to run without a GPU."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "cvllm.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(cvllm_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from compactor_vllm_amd import _lib

    syms = _declared_symbols()
    assert len(syms) >= 20
    L = _lib.lib()
    for s in syms:
        assert hasattr(L, s), f"libcvllm_hip.so does not export {s}"
    assert sorted(_lib.SIGNATURES) == syms, "ctypes signature table out of sync with include/cvllm.h"
    assert _lib.missing_symbols() == []
    assert L.cvllm_version() >= 100
    assert L.cvllm_error_string(-3).decode().startswith("workspace")


def test_host_only_entry_points():
    from compactor_vllm_amd import _lib

    L = _lib.lib()
    # header + max(two-kernel fp32 partials, in-launch merge mailboxes): must cover both layouts
    ws = L.cvllm_decode_workspace_bytes(2, 32, 128, 8)
    assert ws >= 256 + (2 * 8 * 32 * 128 + 2 * 8 * 32) * 4
    assert ws >= 256 + 2 * 32 * 8 * (128 + 8 * 8 + 8) * 4 and ws % 4 == 0
    assert L.cvllm_decode_workspace_bytes(0, 32, 128, 8) == 0
    assert L.cvllm_select_workspace_bytes(3, 8, 1000) == 3 * 8 * 4
    assert L.cvllm_leverage_workspace_bytes(100, 8, 48) == 100 * 8 * 48 * 4
    assert L.cvllm_rank_workspace_bytes(2, 8, 512) > 4 * 2 * 512 * 8 * 4
    # argument validation happens before any launch: null pointers -> CVLLM_ERR_ARG, no GPU touched
    assert L.cvllm_decode_attn(None, None, None, None, None, None, None, None, 0, 1, 8, 2, 128, 128, 4, 1.0, 1, 0,
                               None) == -1
    assert L.cvllm_store_all_kv(None, None, 0, 0, 0, 0, None, None, None, None, None, None, 1, 1, 2, 128, 128, 4, 0,
                                None) == -1


def test_num_splits_heuristic_known_answers():
    """Values of the reference's num_splits_heuristic (sparse_decode_kernel.py:169-192), num_sms = 256
    (SURVEY App. C occupancy note) and H100-like 132."""
    from compactor_vllm_amd.attention.sparse_decode_kernel import num_splits_heuristic, plan_internal_splits

    assert num_splits_heuristic(8, 16384, 256, 12) == 9
    assert num_splits_heuristic(8, 32768, 256, 12) == 9
    assert num_splits_heuristic(64, 65536, 256, 12) == 3
    assert num_splits_heuristic(205, 65536, 256, 12) == 1
    assert num_splits_heuristic(8, 1024, 256, 12) == 1
    assert num_splits_heuristic(1, 65536, 132, 128) == 96
    # internal plan: one workgroup per CU, >= 256 rows per split, honours the caller's hint as a lower bound
    assert plan_internal_splits(8, 40960, None) == 32
    assert plan_internal_splits(8, 40960, 9) == 32
    assert plan_internal_splits(512, 40960, None) == 1
    assert plan_internal_splits(8, 512, None) == 2
    assert plan_internal_splits(8, 512, 12) == 12


def test_split_into_chunks_matches_reference_docstring():
    from compactor_vllm_amd.compression.compactor import split_into_chunks

    assert split_into_chunks([257, 127], 128) == ([256, 1, 127], [128, 128, 1, 127])  # compactor.py:84-87
    assert split_into_chunks([512, 5], 512) == ([512, 5], [512, 5])
    assert split_into_chunks([0, 3], 4) == ([3], [3])


def test_protected_ranges_python_slice_semantics():
    from compactor_vllm_amd.compression.compactor import _protected_ranges
    from oracle import ref_cpu as O

    lens, first, last = [40, 200, 20], [16, 16, 16], [64, 64, 64]
    N = sum(lens)
    ref = torch.zeros(N, 2)
    O.fill_protected(ref, lens, first, last)
    mine = torch.zeros(N, 2)
    for lo, hi in _protected_ranges(lens, first, last, N):
        mine[lo:hi] = float("inf")
    assert torch.equal(ref, mine)


def test_retain_count_bankers_round():
    from oracle import ref_cpu as O

    assert O.retain_count(0.5, 32768, 16, 64, 8) == 130752
    assert O.retain_count(1.0, 50, 16, 64, 8) == 1  # quirk Q2: first+last >= L
    assert O.retain_count(0.5, 81, 0, 0, 1) == 40   # round(40.5) = 40 (banker's)


def test_paged_kv_cache_allocator_cpu():
    from compactor_vllm_amd.kv_cache.page_table import KVAllocationStatus, PagedKVCache

    c = PagedKVCache(num_layers=2, max_logical_pages_per_head=4, num_pages=20, page_size=128, H_kv=2, head_dim=64,
                     max_num_batches=3, dtype=torch.float16, device="cpu")
    assert c.kv_cache.shape == (2, 2, 20 * 128, 64) and c.page_table.shape == (2, 4, 2, 4)
    b = c.new_batch()
    assert b == 1  # row 0 is RESERVED_BATCH
    assert c.reserve_tokens(b, 300) == KVAllocationStatus.SUCCESS
    assert (c.bh_num_pages[:, b] == 3).all()
    assert sorted(c.page_table[0, b].reshape(-1)[[0, 1, 2, 4, 5, 6]].tolist()) == [0, 1, 2, 3, 4, 5]
    assert c.reserve_tokens(b, 300) == KVAllocationStatus.SUCCESS  # already covered
    assert c.reserve_tokens(b, 600) == KVAllocationStatus.EXCEEDS_MAX_SEQUENCE_LENGTH
    b2 = c.new_batch()
    assert c.reserve_tokens(b2, 512) == KVAllocationStatus.SUCCESS
    b3 = c.new_batch()
    assert c.reserve_tokens(b3, 512) == KVAllocationStatus.EXCEEDS_CURRENTLY_AVAILABLE_PAGES
    c.bh_seq_lens[:, b] = 100  # pretend compression kept 100 rows per head
    freed = c.reclaim_pages(b, future_reserve_tokens=20)
    assert freed == 2 * 2 * 2 * (128 * 64 * 2) * 2  # 2 layers x 2 heads x 2 pages, K+V
    assert (c.bh_num_pages[:, b] == 1).all()
    c.free_batch(b)
    assert len(c.free_pages[0]) == 20 - 8
    k, v, pt, bh = c.layer_slices(1)
    assert k.data_ptr() == c.kv_cache[0, 1].data_ptr() and bh.shape == (4, 2)


def test_product_path_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from compactor_vllm_amd.attention.sparse_decode_kernel import head_sparse_decode_attention

    q = torch.zeros(1, 8, 128, dtype=torch.float16)
    kc = torch.zeros(128, 128, dtype=torch.float16)
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        head_sparse_decode_attention(q, kc, kc, torch.ones(1, 2, dtype=torch.int32),
                                     torch.zeros(2, 2, 1, dtype=torch.int32), torch.ones(1, dtype=torch.int32), 2, 128)


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "compactor-vllm_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("# the oracle", ""), f"{f} mentions the oracle"


def test_context_records_and_setters():
    """`Context` / `CompressionContext` keep the reference's attribute names and defaults (utils/context.py:9-52 there);
    set_context is keyword-only, replaces the whole record and rejects unknown names."""
    from compactor_vllm_amd.compression import COMPRESSION_REGISTRY, CompressionMethod
    from compactor_vllm_amd.config.engine_config import AttentionBackend
    from compactor_vllm_amd.utils import context as C

    assert [m.name for m in CompressionMethod] == ["COMPACTOR", "SNAPKV", "NONE"]
    assert [m.value for m in CompressionMethod] == [1, 2, 3]
    assert set(COMPRESSION_REGISTRY) == set(CompressionMethod)
    cc = C.CompressionContext()
    assert (cc.compression_method, cc.compression_chunk_size, cc.max_tokens_to_retain) == (CompressionMethod.COMPACTOR,
                                                                                           -1, 0)
    assert cc.batch_tokens_to_retain is None and cc.PHI is None and cc.context_lens is None
    assert cc.protected_first_tokens is None and cc.protected_last_tokens is None
    C.reset_context()
    d = C.get_context()
    assert (d.is_prefill, d.do_compression, d.max_seqlen_q, d.max_seqlen_k, d.max_bh_len) == (False, False, 0, 0, 0)
    assert d.attention_backend == AttentionBackend.COMPACTOR_TRITON and d.STORE_STREAM is None and d.key_split is None
    bm = torch.ones(2, dtype=torch.int32)
    C.set_context(is_prefill=True, do_compression=True, batch_mapping=bm, max_seqlen_q=7, compression_context=cc)
    e = C.get_context()
    assert e.is_prefill and e.do_compression and e.batch_mapping is bm and e.max_seqlen_q == 7
    assert e.compression_context is cc and e.cu_seqlens_q is None
    C.set_context(is_prefill=False)  # a NEW record: nothing carries over
    f = C.get_context()
    assert not f.is_prefill and f.batch_mapping is None and f.max_seqlen_q == 0
    with pytest.raises(TypeError):
        C.set_context(is_prefill=False, no_such_field=1)
    with pytest.raises(TypeError):
        C.set_context(True)  # keyword-only
    C.reset_context()


def test_compression_params_defaults_and_snapkv_rule():
    from compactor_vllm_amd.compression import BatchCompressionParams, CompressionMethod, SequenceCompressionParams
    from compactor_vllm_amd.config.sampling_params import SamplingParams

    s = SequenceCompressionParams()
    assert (s.compression_ratio, s.protected_first_tokens, s.protected_last_tokens) == (1.0, 16, 64)
    b = BatchCompressionParams()
    assert (b.compression_method, b.do_chunked_compression, b.chunk_size) == (CompressionMethod.COMPACTOR, True, 512)
    assert BatchCompressionParams(compression_method=CompressionMethod.SNAPKV).do_chunked_compression is False
    p = SamplingParams()
    assert (p.temperature, p.max_new_tokens) == (1.0, 256)
    with pytest.raises(ValueError):
        SamplingParams(temperature=-0.1)


def test_register_audit_flags_compiler_use_of_asm_owned_registers(tmp_path):
    """tools/audit_acc_regs.py is what keeps hipcc out of the registers the 4-wave prefill kernel owns through inline
    asm (a[128:255]: output accumulators, v[222:255]: softmax state).  Synthetic assembly: clean, a compiler-emitted
    instruction on an owned AGPR, one on an owned VGPR tuple, and scratch use - the first passes, the others fail; the same
    names inside an ;;#ASMSTART / ;;#ASMEND pair are the kernel's own and pass."""
    import importlib.util
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("audit_acc_regs", os.path.join(root, "tools", "audit_acc_regs.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)

    def asm(body, scratch=0):
        name = "_ZN5cvllm22prefill_attn_w4_kernelINS_4BF16ELi4EEEvPKt"
        return (f"{name}:\n{body}\n.Lfunc_end0:\n"
                f".amdhsa_kernel {name}\n\t.amdhsa_private_segment_fixed_size {scratch}\n.end_amdhsa_kernel\n")

    own = "\t;;#ASMSTART\n\tv_mfma_f32_32x32x16_bf16 a[128:143], v[4:7], v[232:235], a[128:143]\n\tv_add_f32 v224, v224, v228\n\t;;#ASMEND\n"
    cases = {
        "clean": (asm(own + "\tv_mfma_f32_32x32x16_bf16 v[50:65], v[80:83], v[100:103], v[50:65]\n\tv_add_u32_e32 v221, v1, v2\n"), 0),
        "agpr": (asm(own + "\tv_accvgpr_read_b32 v3, a130\n"), 1),
        "vgpr": (asm(own + "\tds_read_b128 v[220:223], v9\n"), 1),
        "scratch": (asm(own, scratch=16), 1),
    }
    for key, (text, want) in cases.items():
        p = tmp_path / f"{key}.s"
        p.write_text(text)
        assert len(mod.audit(str(p))) == want, key
