"""GPU side of the clip-parallel path: the gatherer on CUDA tensors (one rank), and - where the box has two GPUs - two RCCL
ranks through `gather_to_rank0`, `merge_clip_results` and the overlapped per-clip gatherer."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def test_gatherer_single_rank_cuda(cuda):
    from slotvps_amd import parallel
    tmpl = parallel.clip_result_template(2, 8, 16, cuda)
    gat = parallel.ClipResultGatherer(tmpl, depth=2)
    for i in range(3):
        d = gat.submit({k: torch.full_like(v, i + 1) for k, v in tmpl.items()})
    gat.drain()
    torch.cuda.synchronize()
    assert int(gat.last(d)["panoptic_outputs"][0][0, 0, 0]) == 3


def _rccl_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from slotvps_amd import parallel
    r, lr, w = parallel.init_distributed(backend="nccl")
    dev = torch.device("cuda", lr)
    block = torch.full((2, 4, 8), rank + 1, dtype=torch.uint8, device=dev)
    gathered = parallel.gather_to_rank0(block)
    tmpl = parallel.clip_result_template(2, 8, 16, dev)
    gat = parallel.ClipResultGatherer(tmpl, depth=2)
    for i in range(4):
        d = gat.submit({k: torch.full_like(v, 10 * rank + i) for k, v in tmpl.items()})
    gat.drain()
    torch.cuda.synchronize(dev)
    if rank == 0:
        merged = parallel.merge_clip_results(gathered, 4, world)
        got = gat.last(d)
        q.put(([int(m[0, 0]) for m in merged], [int(got["fcn_outputs"][r][0, 0, 0]) for r in range(world)]))
    parallel.barrier()
    torch.distributed.destroy_process_group()


def _one_rank_worker(port, q):
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from slotvps_amd import parallel
    r, lr, w = parallel.init_distributed(backend="nccl", single_rank_group=True)       # init_process_group('nccl', world_size=1, device_id=cuda:0)
    assert dist.is_initialized() and dist.get_backend() == "nccl" and (r, lr, w) == (0, 0, 1)
    dev = torch.device("cuda", lr)
    tmpl = parallel.clip_result_template(5, 1024, 2048, dev)                             # SURVEY 8e's payload: 2 x 10.5 MB uint8 maps + segments per T = 5 clip
    gat = parallel.ClipResultGatherer(tmpl, depth=2)
    assert gat.collective and gat.world == 1 and gat.recv is not None
    compute = torch.zeros((1 << 20,), device=dev)
    seen = []
    for i in range(3):                                                                     # three submits on a depth-2 ring: staging set 0 is reused
        d = gat.submit({k: torch.full_like(v, i + 1) for k, v in tmpl.items()})
        compute += 1.0                                                                     # the compute stream goes on while RCCL gathers on the side stream
        seen.append(d)
    gat.drain()
    torch.cuda.synchronize(dev)
    got = gat.last(d)
    t_max = parallel.max_over_ranks(1.25, dev)                                             # all_reduce(MAX) through RCCL
    blk = parallel.gather_to_rank0(torch.full((2, 4, 8), 7, dtype=torch.uint8, device=dev))
    parallel.barrier()
    q.put(dict(staging=seen, pan=int(got["panoptic_outputs"][0][4, 1023, 2047]), nseg=int(got["num_segments"][0][0]), n_recv=len(got["fcn_outputs"]),
               t_max=t_max, blk=int(blk[0][1, 3, 7]), compute=float(compute[0]), bytes=gat.bytes_per_submit))
    dist.destroy_process_group()


def test_one_rank_nccl_group(cuda):
    """The RCCL code path on the one-GPU box: a process group of ONE rank with the nccl (= RCCL) backend bound to cuda:0, the per-clip
    result gatherer with SURVEY 8e's full-size payload (async dist.gather behind an event on the side stream, staging ring reused),
    max_over_ranks, gather_to_rank0, barrier, destroy - in a child process, so that the group does not outlive the test."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_one_rank_worker, args=(port, q))
    p.start()
    r = q.get(timeout=300)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert r["staging"] == [0, 1, 0] and r["pan"] == 3 and r["nseg"] == 3 and r["n_recv"] == 1
    assert r["t_max"] == 1.25 and r["blk"] == 7 and r["compute"] == 3.0 and r["bytes"] >= 2 * 5 * 1024 * 2048


def test_two_rank_rccl_gather(cuda):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    merged, last = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert merged == [1, 2, 1, 2] and last == [3, 13]
