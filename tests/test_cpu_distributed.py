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
