"""Multi-GPU driver helpers: sequences shard across one-process-per-GPU replicas with NO data-path collective
(SURVEY §8(e)).  torch.distributed (RCCL on GPUs, gloo in the CPU tests) is used only for the timing fence, the
max-over-ranks reduction and gathering small per-sequence results."""
from __future__ import annotations

import time
from typing import List, Sequence


def request_cost(prompt_len: int, max_new: int, ratio: float) -> float:
    """Relative cost of one request: prefill attention ~ L^2, decode ~ retained rows x new tokens."""
    return float(prompt_len) ** 2 + ratio * prompt_len * max_new


def partition_lpt(costs: Sequence[float], world: int) -> List[List[int]]:
    """Longest-processing-time-first partition of request indices over `world` replicas (deterministic:
    ties broken by index, so every rank computes the same partition without communicating)."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0.0] * world
    parts: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda j: (loads[j], j))
        parts[r].append(i)
        loads[r] += costs[i]
    return [sorted(p) for p in parts]


def timed_region(fn, dist=None, sync=None):
    """barrier + device sync on both sides, wall time of fn(), MAX over ranks (bench.py contract)."""
    import torch

    def fence():
        if dist is not None:
            dist.barrier()
        if sync is not None:
            sync()

    fence()
    t0 = time.perf_counter()
    out = fn()
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64)
        if sync is not None and torch.cuda.is_available():
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, out


def gather_by_seq_id(local: dict, dist=None) -> dict:
    """Merge {seq_id: result} dictionaries of all ranks on every rank (host objects, tiny)."""
    if dist is None:
        return dict(local)
    bucket = [None] * dist.get_world_size()
    dist.all_gather_object(bucket, local)
    merged = {}
    for d in bucket:
        merged.update(d)
    return merged
