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
