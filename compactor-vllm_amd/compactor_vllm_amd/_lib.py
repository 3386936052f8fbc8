"""ctypes binding of libcvllm_hip.so (the C ABI declared in include/cvllm.h).

There is NO fallback: if the shared library is missing or a call returns a non-zero status the
caller gets a RuntimeError.  torch is used only for device memory and the current HIP stream.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CVLLM_LIB_PATH") or os.path.join(_HERE, "libcvllm_hip.so")  # override: A/B builds

_lib = None

_P, _I, _L, _F, _Z = c_void_p, c_int, c_int64, c_float, c_size_t

# name -> (restype, argtypes); mirrors include/cvllm.h one-to-one
SIGNATURES = {
    "cvllm_version": (_I, []),
    "cvllm_error_string": (c_char_p, [_I]),
    "cvllm_decode_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "cvllm_decode_attn": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _Z, _I, _I, _I, _I, _I, _I, _F, _I, _I, _P]),
    "cvllm_decode_attn_lse": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _I, _I, _I, _I, _I, _I, _F, _I, _I, _P]),
    "cvllm_decode_merge_shards": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "cvllm_decode_append_attn": (_I, [_P, _P, _P, _L, _L, _L, _L, _P, _P, _P, _P, _P, _P, _P, _Z, _I, _I, _I, _I, _I,
                                      _I, _F, _I, _I, _I, _P]),
    "cvllm_decode_merge_status": (_I, [_P, _P]),
    "cvllm_decode_set_merge_mode": (_I, [_I]),
    "cvllm_num_splits": (_I, [_I, _I, _I, _I]),
    "cvllm_store_decode_kv": (_I, [_P, _P, _L, _L, _L, _L, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "cvllm_store_all_kv": (_I, [_P, _P, _L, _L, _L, _L, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "cvllm_prefill_attn": (_I, [_P, _P, _P, _L, _L, _L, _L, _L, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I,
                                _I, _I, _I, _F, _I, _P]),
    "cvllm_zscore_workspace_bytes": (_Z, [_I]),
    "cvllm_zscore_segments": (_I, [_P, _I, _P, _I, _I, _P, _I, _F, _P, _I, _I, _P, _Z, _P]),
    "cvllm_chunk_attn_mass": (_I, [_P, _P, _L, _L, _L, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _I, _P]),
    "cvllm_leverage_workspace_bytes": (_Z, [_I, _I, _I]),
    "cvllm_leverage_scores": (_I, [_P, _L, _L, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _I, _P, _Z, _P]),
    "cvllm_snapkv_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "cvllm_snapkv_scores": (_I, [_P, _P, _L, _L, _L, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _I, _I, _I, _P, _Z, _P]),
    "cvllm_snapkv_scores_wb": (_I, [_P, _P, _L, _L, _L, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _I, _I, _I, _P, _Z, _P]),
    "cvllm_zscore_windowed": (_I, [_P, _I, _P, _P, _I, _I, _I, _F, _I, _P, _Z, _P]),
    "cvllm_select_workspace_bytes": (_Z, [_I, _I, _I]),
    "cvllm_select_topk": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "cvllm_select_status": (_I, [_P]),
    "cvllm_compact_store": (_I, [_P, _P, _L, _L, _L, _L, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I,
                                 _I, _I, _P]),
    "cvllm_compact_cache_inplace": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "cvllm_store_topk_ranked": (_I, [_P, _P, _L, _L, _L, _L, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I,
                                     _I, _I, _I, _I, _I, _P]),
    "cvllm_qkv_rope_producer": (_I, [_P, _L, _P, _P, _P, _P, _F, _P, _L, _P, _L, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I,
                                     _I, _I, _I, _I, _I, _P]),
    "cvllm_rank_workspace_bytes": (_Z, [_I, _I, _I]),
    "cvllm_rank_indices": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _Z, _P]),
}


def lib() -> ctypes.CDLL:
    """Load (once) and return the HIP library; raise loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build the HIP extension first "
                "(python compactor-vllm_amd/build.py or __graft_entry__.build()). "
                "There is no CPU / PyTorch fallback for the compactor hot path."
            )
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name, None)
            if fn is None:  # header/library mismatch: calling it later raises AttributeError loudly
                continue
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def missing_symbols() -> list:
    """Entry points declared in include/cvllm.h that the built library does not export."""
    L = lib()
    return [name for name in SIGNATURES if not hasattr(L, name)]


def check(status: int, what: str) -> None:
    if status != 0:
        msg = lib().cvllm_error_string(status).decode()
        raise RuntimeError(f"{what} failed: {msg} (status {status})")


def ptr(t: torch.Tensor | None):
    return None if t is None else t.data_ptr()


def stream() -> int:
    """The CURRENT torch stream (the reference switches it with `with STORE_STREAM:`)."""
    return torch.cuda.current_stream().cuda_stream


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float16:
        return 0
    if dt == torch.bfloat16:
        return 1
    raise TypeError(f"compactor_vllm_amd kernels support float16 / bfloat16, got {dt}")


def score_dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return 2
    return dtype_code(dt)


def require_cuda(*tensors: torch.Tensor) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "compactor_vllm_amd: tensors must live on the GPU (the hot path has no CPU implementation)"
            )


def i32(t: torch.Tensor) -> torch.Tensor:
    t = t if t.dtype == torch.int32 else t.to(torch.int32)
    return t if t.is_contiguous() else t.contiguous()
