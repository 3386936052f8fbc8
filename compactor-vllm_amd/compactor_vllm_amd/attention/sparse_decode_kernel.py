"""Decode-time head-sparse attention over the paged KV cache (MI355X HIP kernel).

Mirror of the reference module `compactor_vllm/attention/sparse_decode_kernel.py`
(wrapper :10-165, heuristic :169-192): same function names, argument order, defaults and
assertions; the Triton stage-1 / stage-2 kernels are replaced by `cvllm_decode_attn`.
"""
from __future__ import annotations

import functools
import math

import torch

from .. import _lib

# MI355X: 256 CUs.  The ring kernel keeps ~128 KiB of LDS per workgroup, i.e. one workgroup per CU, and a CU
# streams at most ~24 GB/s, so the plan is one workgroup per CU; each streams a contiguous run of rows of
# one (batch, kv-head).
_TARGET_WORKGROUPS = 256  # MI355X; replaced by the device's CU count on first use (_cus)
_MIN_ROWS_PER_SPLIT = 256
_MAX_INTERNAL_SPLITS = 128

_workspaces: dict = {}
_retired: list = []  # outgrown workspaces: a HIP graph captured earlier may still hold their address


def _workspace(device: torch.device, nbytes: int) -> torch.Tensor:
    """Persistent per-(device, stream) scratch so decode stays graph-capture safe (no allocation inside a captured
    region as long as an un-captured call of at least that size ran on the same stream before).

    ZERO-initialised, and every completed call leaves it all zeros again: the in-launch split merge uses it as
    mailboxes in which 0 means "not written yet" (csrc/decode_attn.hip).  A buffer that has to grow is replaced, never
    freed, because a captured graph replays with the address it was captured with."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            _retired.append(buf)
        buf = torch.zeros(max(nbytes, 8 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf


_MERGE_MODES = {"default": 0, "two-kernel": 1, "in-launch": 2}


def set_merge_mode(mode: str) -> str:
    """Process-wide choice of the split merge for grids that fit the chip: "in-launch" (one launch, the default; needs
    every workgroup co-resident: poll `merge_status`), "two-kernel" (fp32 partials + decode_stage2_kernel), "default" (the
    environment's CVLLM_DECODE_MERGE, else in-launch).  Returns the previous setting.  A captured graph keeps the mode it
    was captured with."""
    prev = int(_lib.lib().cvllm_decode_set_merge_mode(_MERGE_MODES[mode]))
    return {v: k for k, v in _MERGE_MODES.items()}[prev]


def merge_status(device: torch.device | None = None) -> int:
    """Health check of the in-launch split merge over EVERY live decode
    workspace of the device - graphs launch on their capture stream's workspace, eager calls on the current stream's, so
    looking at one stream's buffer would miss the others.  Synchronises the device.  0 = fine; 1 = some call gave up
    waiting for a sibling split (its output rows hold NaN) - the affected workspaces are re-zeroed here so that later
    calls start clean."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    bufs = [buf for (dev_index, _), buf in _workspaces.items() if dev_index == device.index]
    if not bufs:
        return 0
    torch.cuda.synchronize(device)
    bad = 0
    for buf in bufs:
        st = int(_lib.lib().cvllm_decode_merge_status(buf.data_ptr(), _lib.stream()))
        if st < 0:
            _lib.check(st, "cvllm_decode_merge_status")
        if st:
            buf.zero_()
            bad = 1
    return bad


# MI355X: one ring workgroup per CU streams at the chip rate, so the split count is the largest that keeps
# batch * kv-heads * splits within the chip's CUs
@functools.lru_cache(maxsize=None)
def _cus(device_index: int) -> int:
    global _TARGET_WORKGROUPS
    _TARGET_WORKGROUPS = int(torch.cuda.get_device_properties(device_index).multi_processor_count)
    return _TARGET_WORKGROUPS


def plan_internal_splits(n_bh: int, max_len_bound: int, key_split: int | None) -> int:
    """Number of key splits the HIP kernel uses.  Depends only on host integers (batch, heads,
    the page-table width bound, the caller's key_split hint) so the launch is capture safe."""
    want = max(1, _TARGET_WORKGROUPS // max(n_bh, 1))
    cap = max(1, max_len_bound // _MIN_ROWS_PER_SPLIT)
    s = min(want, cap, _MAX_INTERNAL_SPLITS)
    if key_split:
        s = max(s, min(int(key_split), _MAX_INTERNAL_SPLITS))
    return max(1, s)


def head_sparse_decode_attention(
    q: torch.Tensor,
    k: torch.Tensor,
    v: torch.Tensor,
    seq_lens_bh: torch.Tensor,
    global_page_table: torch.Tensor,
    batch_mapping: torch.Tensor,
    HKV: int,
    PAGE_SIZE: int,
    sm_scale: float = None,
    key_split: int = None,
    *,
    return_lse: bool = False,
):
    """Same contract as the reference wrapper (sparse_decode_kernel.py:10-68).  `return_lse=True` (extension, used by
    attention/cross_gpu_decode.py) also returns lse [B, HQ] fp32: the natural-log LSE of the scaled logits.

    q: [B, HQ, D] (or [B, HQ, 1, D]); k, v: global caches [CACHE_SIZE, D]; seq_lens_bh [B, HKV]
    int32 (lengths including the current token); global_page_table [MAX_BATCHES, HKV, P];
    batch_mapping [B].  Returns [B, HQ, D] in q.dtype.

    Differences, all documented in DESIGN.md: `key_split` is a lower-bound hint — the HIP kernel
    sub-splits further to fill 256 CUs (results equal up to fp32 rounding); with key_split=None no
    host sync happens (the reference reads seq_lens_bh.max()); rows with L == 0 return zeros.
    """
    _lib.require_cuda(q, k, v, seq_lens_bh, global_page_table, batch_mapping)
    if q.ndim != 3:
        assert q.ndim == 4
        B, HQ, S, D = q.shape
        assert S == 1, "head_sparse_decode_attention only supports q_len=1"
        q = q.squeeze(-2)
    B, HQ, D = q.shape
    assert PAGE_SIZE % 32 == 0, "PAGE_SIZE must be divisible by 32"
    GROUP_M = HQ // HKV
    assert GROUP_M * HKV == HQ, "HQ must be divisible by H_kv"
    assert B <= 32767, "too many batches"
    assert global_page_table.shape[1] == HKV
    assert q.is_contiguous()
    assert (D & (D - 1)) == 0, "D must be a power of 2"
    assert k.is_contiguous() and v.is_contiguous() and k.shape[1] == D and v.shape[1] == D
    assert global_page_table.is_contiguous()
    n_lp = global_page_table.shape[-1]
    sm_scale = 1 / math.sqrt(D) if sm_scale is None else sm_scale

    seq_lens_bh = _lib.i32(seq_lens_bh)
    batch_mapping = _lib.i32(batch_mapping)
    page_table = _lib.i32(global_page_table)
    _cus(q.device.index)
    n_splits = plan_internal_splits(B * HKV, n_lp * PAGE_SIZE, key_split)

    L = _lib.lib()
    out = torch.empty_like(q)
    ws = None
    ws_bytes = 0
    if n_splits > 1:
        ws_bytes = L.cvllm_decode_workspace_bytes(B, HQ, D, n_splits)
        ws = _workspace(q.device, ws_bytes)
    if return_lse:
        lse = torch.empty((B, HQ), dtype=torch.float32, device=q.device)
        st = L.cvllm_decode_attn_lse(
            q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), lse.data_ptr(), seq_lens_bh.data_ptr(),
            page_table.data_ptr(), batch_mapping.data_ptr(), _lib.ptr(ws), ws_bytes, B, HQ, HKV, D, PAGE_SIZE,
            n_lp, float(sm_scale), n_splits, _lib.dtype_code(q.dtype), _lib.stream(),
        )
        _lib.check(st, "cvllm_decode_attn_lse")
        return out, lse
    st = L.cvllm_decode_attn(
        q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), seq_lens_bh.data_ptr(),
        page_table.data_ptr(), batch_mapping.data_ptr(), _lib.ptr(ws), ws_bytes, B, HQ, HKV, D, PAGE_SIZE,
        n_lp, float(sm_scale), n_splits, _lib.dtype_code(q.dtype), _lib.stream(),
    )
    _lib.check(st, "cvllm_decode_attn")
    return out


def fused_decode_step(
    q: torch.Tensor,  # [B, HQ, D]
    key: torch.Tensor,  # [B, HKV, D] new token
    value: torch.Tensor,
    k_cache: torch.Tensor,
    v_cache: torch.Tensor,
    bh_seq_lens_layer: torch.Tensor,  # [Bmax+1, HKV] int32, the layer's table, UPDATED in place
    page_table: torch.Tensor,
    batch_mapping: torch.Tensor,  # [B] int32
    HKV: int,
    PAGE_SIZE: int,
    sm_scale: float = None,
    key_split: int = None,
    reserved_batch: int = 0,
    max_len_hint: int = 0,
):
    """decode_store_kv + head_sparse_decode_attention + length write-back of the reference's decode branch
    (layers/attention.py:127-160) in one C-ABI call that works directly on the layer's length table."""
    _lib.require_cuda(q, key, value, k_cache, v_cache, bh_seq_lens_layer, page_table, batch_mapping)
    B, HQ, D = q.shape
    assert key.shape == (B, HKV, D) and value.shape == (B, HKV, D)
    if not q.is_contiguous():
        q = q.contiguous()  # e.g. a view of a fused [B, HQ+HKV, D] RoPE output
    assert key.stride(-1) == 1 and value.stride(-1) == 1
    assert PAGE_SIZE % 32 == 0 and HQ % HKV == 0
    assert bh_seq_lens_layer.is_contiguous() and bh_seq_lens_layer.dtype == torch.int32
    assert batch_mapping.dtype == torch.int32 and page_table.is_contiguous() and page_table.dtype == torch.int32
    n_lp = page_table.shape[-1]
    sm_scale = 1 / math.sqrt(D) if sm_scale is None else sm_scale
    _cus(q.device.index)
    bound = n_lp * PAGE_SIZE if not max_len_hint else min(n_lp * PAGE_SIZE, int(max_len_hint))  # tuning only
    n_splits = plan_internal_splits(B * HKV, bound, key_split)
    L = _lib.lib()
    out = torch.empty_like(q)
    ws, ws_bytes = None, 0
    if n_splits > 1:
        ws_bytes = L.cvllm_decode_workspace_bytes(B, HQ, D, n_splits)
        ws = _workspace(q.device, ws_bytes)
    st = L.cvllm_decode_append_attn(
        q.data_ptr(), key.data_ptr(), value.data_ptr(), key.stride(0), key.stride(1), value.stride(0), value.stride(1),
        k_cache.data_ptr(), v_cache.data_ptr(), out.data_ptr(), bh_seq_lens_layer.data_ptr(), page_table.data_ptr(),
        batch_mapping.data_ptr(), _lib.ptr(ws), ws_bytes, B, HQ, HKV, D, PAGE_SIZE, n_lp, float(sm_scale), n_splits,
        int(reserved_batch), _lib.dtype_code(q.dtype), _lib.stream(),
    )
    _lib.check(st, "cvllm_decode_append_attn")
    return out


@functools.lru_cache(maxsize=128)
def num_splits_heuristic(total_mblocks: int, max_seq_len: int, num_sms: int, max_splits: int) -> int:
    """Reference heuristic (sparse_decode_kernel.py:169-192), evaluated by the C ABI's host
    restatement so both sides agree bit for bit; kept because ModelRunner imports it."""
    return int(_lib.lib().cvllm_num_splits(int(total_mblocks), int(max_seq_len), int(num_sms), int(max_splits)))
