"""world_size-2 gloo test of the replica driver: deterministic sharding, max-over-ranks timing, result gather."""
import os
import socket
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bench_dist import gather_by_seq_id, partition_lpt, request_cost, timed_region


def test_partition_lpt_is_balanced_and_deterministic():
    lens = [131072, 4096, 32768, 32768, 16384, 8192, 65536, 1024]
    costs = [request_cost(L, 256, 0.5) for L in lens]
    parts = partition_lpt(costs, 2)
    assert sorted(parts[0] + parts[1]) == list(range(8))
    assert parts == partition_lpt(costs, 2)
    loads = [sum(costs[i] for i in p) for p in parts]
    assert max(loads) <= sum(costs) * 0.5 + max(costs)
    assert partition_lpt(costs, 1) == [list(range(8))]
    eight = partition_lpt([request_cost(131072, 256, 0.5)] * 64, 8)
    assert all(len(p) == 8 for p in eight)  # BASELINE config 5: 64 sequences -> 8 per GPU


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lens = [300, 100, 200, 50, 250]
    parts = partition_lpt([request_cost(L, 8, 0.5) for L in lens], world)
    mine = parts[rank]

    def work():
        time.sleep(0.05 * (rank + 1))  # rank 1 is slower: the reported time must be ITS time
        return {i: lens[i] * 2 for i in mine}

    dt, local = timed_region(work, dist=dist)
    merged = gather_by_seq_id(local, dist)
    tot = torch.tensor([float(sum(lens[i] for i in mine))])
    dist.all_reduce(tot)
    q.put((rank, dt, merged, float(tot.item()), mine))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_replicas_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, dt0, m0, tot0, mine0), (r1, dt1, m1, tot1, mine1) = res
    assert abs(dt0 - dt1) < 1e-9 and dt0 >= 0.1  # MAX over ranks, identical on both
    assert m0 == m1 == {0: 600, 1: 200, 2: 400, 3: 100, 4: 500}
    assert tot0 == tot1 == 900.0
    assert sorted(mine0 + mine1) == [0, 1, 2, 3, 4] and not set(mine0) & set(mine1)


# ------------------------------------------------------------------------------------------ f-4
def _split_kv_worker(rank, world, port, q):
    """Each rank holds every other 128-row block of a 3-sequence cache; the all-gather / record layout / merge order of
    attention/cross_gpu_decode.py is exercised over gloo with the two HIP calls replaced by the CPU oracle."""
    import math
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from compactor_vllm_amd.attention.cross_gpu_decode import split_kv_decode_attention
    from helpers import mk_paged
    from oracle import ref_cpu as O

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, HQ, HKV, D, PS = 3, 8, 2, 64, 128
    g = torch.Generator().manual_seed(0)  # the same data on every rank
    lens = torch.tensor([[700, 300], [129, 0], [1, 513]], dtype=torch.int32)
    kc, vc, pt, bm, P = mk_paged(B, HKV, D, PS, lens, torch.float16, seed=5)
    qv = torch.randn(B, HQ, D, generator=g).to(torch.float16)
    # this rank's slice: pages rank, rank + world, ... of every (b, h), re-packed into a local page table
    my_pt = torch.zeros_like(pt)
    my_lens = torch.zeros_like(lens)
    for b in range(B):
        for h in range(HKV):
            L = int(lens[b, h])
            pages = [p for p in range(-(-L // PS)) if p % world == rank]
            for j, p in enumerate(pages):
                my_pt[int(bm[b]), h, j] = pt[int(bm[b]), h, p]
            # rows held: whole pages, except the sequence's last (partial) page
            held = 0
            for p in pages:
                held += min(PS, L - p * PS)
            my_lens[b, h] = held
    # a partial LAST page in the middle of a local list would break the "rows 0..len-1" layout: only the final page of a
    # sequence is partial, and it is the last entry of its owner's list
    scale = 1.0 / math.sqrt(D)
    out = split_kv_decode_attention(
        qv, None, None, None, None, None, HKV, PS, group=None, sm_scale=scale,
        local_attention=lambda qq: O.decode_attention_lse(qq, kc, vc, my_lens, my_pt, bm, HKV, PS, scale),
        merge=O.merge_shards)
    ref = O.decode_attention(qv, kc, vc, lens, pt, bm, HKV, PS, scale)
    q.put((rank, float((out.float() - ref.float()).abs().max()), my_lens.sum().item()))
    dist.barrier()
    dist.destroy_process_group()


def test_split_kv_decode_protocol_two_ranks_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_split_kv_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(err < 2e-3 for _, err, _ in res), res  # every rank ends with the full-attention result
    assert res[0][2] + res[1][2] == 700 + 300 + 129 + 1 + 513  # the two slices partition the rows
