"""CPU: the engine's host policy (f-1) against traces of the REFERENCE's own classes.

tests/golden/engine_policy.json was written by tests/golden/gen_fixtures.py `engine_policy` from the reference's
PagedKVCache (kv_cache/page_table.py:144-291) and Scheduler.get_prefill_batch (core/scheduler.py:65-108) run on scripted
operation lists: every operation's return value and the state it leaves (free rows, free pages per layer, pages held and
page-table entries of every live row).  Here the same scripts run on kv_cache/page_table.py and core/scheduler.py of this
repository.
"""
import json
import os
from types import SimpleNamespace

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def policy():
    with open(os.path.join(HERE, "golden", "engine_policy.json")) as f:
        return json.load(f)


def test_page_allocator_equals_reference_trace(policy):
    from compactor_vllm_amd.kv_cache.page_table import PagedKVCache

    cfg = policy["allocator"]["config"]
    cache = PagedKVCache(head_dim=8, dtype=torch.float16, device="cpu", **cfg)
    L, H = cfg["num_layers"], cfg["H_kv"]
    rows = {}
    for i, step in enumerate(policy["allocator"]["ops"]):
        op, a, want = step["op"], step["args"], step["after"]
        if op == "new_batch":
            ret = cache.new_batch()
            if ret is not None:
                rows[a[0]] = int(ret)
        elif op == "reserve":
            ret = cache.reserve_tokens(rows[a[0]], a[1]).name
        elif op == "set_lens":
            cache.bh_seq_lens[:, rows[a[0]]] = torch.tensor(a[1], dtype=torch.int32)
            ret = None
        elif op == "reclaim":
            ret = cache.reclaim_pages(rows[a[0]], a[1])
        else:
            ret = cache.free_batch(rows.pop(a[0]))
        where = f"operation {i}: {op}{tuple(a)}"
        assert ret == want["ret"], where
        assert list(cache.free_batches) == want["free_batches"], where
        assert [sorted(pool.free) for pool in cache.free_pages] == want["free_pages"], where
        assert cache.bh_num_pages.tolist() == want["num_pages"], where
        assert cache.bh_seq_lens.tolist() == want["lens"], where
        for name, r in rows.items():
            npg = cache.bh_num_pages[:, r]
            got = [[cache.page_table[l, r, h, : int(npg[l, h])].tolist() for h in range(H)] for l in range(L)]
            assert got == want["tables"][name], where
    assert not rows


def test_prefill_admission_equals_reference_trace(policy):
    from compactor_vllm_amd.config.sampling_params import SamplingParams
    from compactor_vllm_amd.core.scheduler import Scheduler
    from compactor_vllm_amd.utils.sequence import Sequence

    for case in policy["scheduler"]:
        seqs = [Sequence(prompt_token_ids=[1] * pl, sampling_params=SamplingParams(max_new_tokens=mn))
                for pl, mn in case["prompts"]]
        mgr = SimpleNamespace(chunked_prefill=False, **case["manager"])
        sc = Scheduler(seqs, mgr)
        rounds = []
        for _ in range(4):
            batch = sc.get_prefill_batch()
            rounds.append([seqs.index(s) for s in batch])
            if not batch:
                break
            sc.add_running_sequence_ids([s.seq_id for s in batch], update_status=True)
        assert rounds == case["rounds"], case["name"]
        assert sc.total_tokens_input == case["total_tokens_input"], case["name"]
