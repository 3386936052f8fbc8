"""Loader for the upstream reference (vnchari/compactor-vllm) under the Triton CPU interpreter.

THIS CONTAINER ONLY: /root/reference does not exist on the GPU box.  Used exclusively by
tests/golden/gen_fixtures.py to produce the committed golden vectors; nothing in the product
path, the `-m gpu` tests, smoke() or bench.py imports this module.

Run with:  PYTHONDONTWRITEBYTECODE=1 TRITON_INTERPRET=1 python tests/golden/gen_fixtures.py

Harness-side pins (the reference files themselves are never modified):
  * a bare namespace package `compactor_vllm` so the top-level __init__ (which needs flash_attn
    and a tokenizer stack) is skipped;
  * one fixed Triton config per autotuned kernel (the autotuner cannot benchmark on CPU);
  * torch.cuda.device / Tensor.cuda / linalg.svd(driver=) made CPU-neutral; for the host-policy vectors
    (`load_engine_policy`) also the hard-coded device="cuda" of torch.tensor / torch.empty / torch.as_tensor
    (kv_cache/page_table.py:185, write_page_table.py:42, core/memory_manager.py:59);
  * bf16 tl.dot up-converted to fp32 in the interpreter (it otherwise multiplies raw uint16 bits).
"""
import contextlib
import importlib.machinery
import os
import sys
import types

assert os.environ.get("TRITON_INTERPRET") == "1", "run with TRITON_INTERPRET=1"
sys.dont_write_bytecode = True

import numpy as np
import torch
import triton
import triton.language as tl

SRC = "/root/reference/src/compactor_vllm"


def load():
    pkg = types.ModuleType("compactor_vllm")
    pkg.__path__ = [SRC]
    pkg.__spec__ = importlib.machinery.ModuleSpec("compactor_vllm", None, is_package=True)
    pkg.__spec__.submodule_search_locations = [SRC]
    sys.modules["compactor_vllm"] = pkg

    torch.cuda.device = lambda d: contextlib.nullcontext()
    torch.Tensor.cuda = lambda self, *a, **k: self
    _svd = torch.linalg.svd
    torch.linalg.svd = lambda A, full_matrices=True, driver=None: _svd(A, full_matrices=full_matrices)

    import compactor_vllm.attention.sparse_decode_kernel as dk
    import compactor_vllm.attention.sparse_varlen_kernel as pk
    import compactor_vllm.compression.common as cm
    import compactor_vllm.compression.compactor as cp
    import compactor_vllm.compression.snapkv as sk
    import compactor_vllm.kv_cache.store_kv_cache as st

    pk._causal_head_sparse_varlen_with_cache.configs = [
        triton.Config({"BLOCK_N": 64, "BLOCK_M": 64, "WARPSPEC": False}, num_warps=4, num_stages=3)
    ]
    dk._varkv_stage1_groupM.configs = [
        triton.Config(
            {"BLOCK_N": 64, "MIN_BLOCK_KV": 8, "WARPSPEC": False},
            num_warps=4,
            num_stages=2,
            pre_hook=dk._stage1_host_desc_pre_hook,
        )
    ]
    cp._zscore_per_batch_epilogue_no_window.configs = [triton.Config({"BLOCK_K": 128})]
    sk._lse_and_store_logits_kernel.configs = [
        triton.Config({"BLOCK_Q": 64, "BLOCK_K": 64}, num_warps=4, num_stages=3)
    ]
    sk._scores_from_logits_kernel.configs = [triton.Config({"BLOCK_Q": 64, "BLOCK_K": 128})]
    sk._zscore_per_batch_epilogue.configs = [triton.Config({"BLOCK_K": 128})]

    from triton.runtime import interpreter as _I

    _orig_dot = _I.InterpreterBuilder.create_dot

    def _dot(self, a, b, d, input_precision, max_num_imprecise_acc):
        def up(h):
            if h.dtype.scalar == tl.bfloat16:
                return _I.TensorHandle(
                    _I._convert_float(h.data, tl.bfloat16, tl.float32, None).view(np.float32), tl.float32
                )
            return h

        return _orig_dot(self, up(a), up(b), d, input_precision, max_num_imprecise_acc)

    _I.InterpreterBuilder.create_dot = _dot
    return types.SimpleNamespace(dk=dk, pk=pk, cm=cm, cp=cp, sk=sk, st=st)


def load_engine_policy():
    """The reference's host-side policy classes (page allocator, prefill admission) for tests/golden/gen_fixtures.py
    `engine_policy`.  Call after load().  Their three hard-coded device="cuda" arguments are redirected to the CPU the
    same way Tensor.cuda is (harness side; the reference files are untouched)."""

    def cpu_device(fn):
        def wrapped(*a, **k):
            if isinstance(k.get("device"), str) and k["device"].startswith("cuda"):
                k["device"] = "cpu"
            return fn(*a, **k)

        return wrapped

    torch.tensor = cpu_device(torch.tensor)
    torch.empty = cpu_device(torch.empty)
    torch.as_tensor = cpu_device(torch.as_tensor)
    import compactor_vllm.core.scheduler as sched
    import compactor_vllm.kv_cache.page_table as ptab
    import compactor_vllm.utils.sequence as seq
    from compactor_vllm.config.sampling_params import SamplingParams

    return types.SimpleNamespace(sched=sched, ptab=ptab, seq=seq, SamplingParams=SamplingParams)
