"""N>1 path on CPU: two gloo ranks shard the ensembles and run the report reductions
bench.py uses (max elapsed, sum frames, min flags).  No data-path collective exists."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dabgpu.shard import ensembles_of_rank, reduce_report


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = ensembles_of_rank(128, world, rank)
    elapsed, frames, flags = reduce_report(dist, torch.device("cpu"), 1.0 + rank, len(mine) * 16,
                                           [True, rank == 0])
    dist.barrier()
    q.put((rank, mine, elapsed, frames, flags))
    dist.destroy_process_group()


def test_two_rank_sharding_and_report():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    owned = res[0][1] + res[1][1]
    assert sorted(owned) == list(range(128)) and not set(res[0][1]) & set(res[1][1])
    for _, _, elapsed, frames, flags in res:
        assert elapsed == 2.0 and frames == 128 * 16 and flags == [True, False]


def test_single_process_report_passthrough():
    assert reduce_report(None, None, 0.5, 10, [True]) == (0.5, 10, [True])
    assert ensembles_of_rank(10, 4, 3) == [3, 7]
