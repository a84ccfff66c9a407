"""Clip-parallel sharding + result gather with two gloo ranks on CPU (the N > 1 path of bench.py)."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_clips, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from slotvps_amd import parallel
    r, lr, w = parallel.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    mine = parallel.clips_of_rank(n_clips, rank, world)
    per_rank = -(-n_clips // world)
    # a clip's "result": uint8 map whose content encodes the clip id
    block = torch.zeros((per_rank, 4, 8), dtype=torch.uint8)
    for j, c in enumerate(mine):
        block[j].fill_(c + 1)
    parallel.barrier()
    gathered = parallel.gather_to_rank0(block)
    tmax = parallel.max_over_ranks(float(rank + 1), torch.device("cpu"))
    assert tmax == float(world)
    if rank == 0:
        merged = parallel.merge_clip_results(gathered, n_clips, world)
        q.put([int(m[0, 0]) for m in merged])
    else:
        assert gathered is None
    parallel.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_gloo_gather():
    world, n_clips = 2, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_clips, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert out == [1, 2, 3, 4, 5]          # clip c came back in position c


def test_single_process_helpers():
    from slotvps_amd import parallel
    assert parallel.clips_of_rank(7, 1, 3) == [1, 4]
    t = torch.arange(6).reshape(2, 3)
    assert parallel.gather_to_rank0(t)[0] is t
    assert parallel.max_over_ranks(3.5, torch.device("cpu")) == 3.5
    merged = parallel.merge_clip_results([torch.tensor([[0], [2]]), torch.tensor([[1], [9]])], 3, 2)
    assert [int(m) for m in merged] == [0, 1, 2]
