"""Clip-parallel execution: one process per GPU, independent clips, no data-path collective.

The reference's inference harness is single-GPU (tools/test_eval_vpq.py:70 sets distributed=False and
wraps the model in MMDataParallel(device_ids=[gpu0]), :134). Videos share nothing (the tracker state is
per video, vps_temporal_slots.py:223-237), so the path shards by clip: clip c -> rank c % world. The
only exchange is the gather of the per-clip results to rank 0 (RCCL over xGMI on GPUs: rank 0 receives
from its 7 peers over 7 distinct links; gloo on CPU for the tests).
"""
import os

import torch
import torch.distributed as dist


def dist_env():
    """(rank, local_rank, world_size) from the torch.distributed.run environment (defaults: single process)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_distributed(backend=None):
    """Initialise torch.distributed when launched with WORLD_SIZE > 1. backend: 'nccl' (= RCCL on ROCm)
    on GPUs, 'gloo' on CPU. Returns (rank, local_rank, world)."""
    rank, local_rank, world = dist_env()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def clips_of_rank(n_clips, rank, world):
    """Round-robin clip -> rank assignment (clip c runs on rank c % world)."""
    return list(range(rank, n_clips, world))


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device):
    """Max of a python float over all ranks (timing: the job is as slow as its slowest rank)."""
    if not (dist.is_available() and dist.is_initialized()):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_to_rank0(tensor):
    """Gather equally shaped per-rank result tensors to rank 0. Returns the list (rank order) on rank 0,
    None elsewhere. Single process: [tensor]."""
    if not (dist.is_available() and dist.is_initialized()):
        return [tensor]
    world, rank = dist.get_world_size(), dist.get_rank()
    tensor = tensor.contiguous()
    out = [torch.empty_like(tensor) for _ in range(world)] if rank == 0 else None
    dist.gather(tensor, gather_list=out, dst=0)
    return out


def merge_clip_results(gathered, n_clips, world):
    """Undo the round-robin sharding: gathered[r][j] is clip r + j * world. Returns a list of n_clips."""
    merged = [None] * n_clips
    for r, block in enumerate(gathered):
        for j in range(block.shape[0]):
            c = r + j * world
            if c < n_clips:
                merged[c] = block[j]
    return merged
