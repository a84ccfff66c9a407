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


def host_cores():
    """Cores this process may actually use: the affinity mask, capped by the cgroup CPU quota where one is set (a box that shows 256
    logical CPUs but grants 16 is 16 cores)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as fh:
                txt = fh.read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota = txt[0]
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh2:
                    period = float(fh2.read())
            if quota not in ("max", "-1"):
                n = max(1, min(n, int(float(quota) / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


class host_pools:
    """Context manager: size the host thread pools (torch intra-op = OpenMP, and the BLAS pool behind numpy) to this rank's share of the
    cores the job may use for the duration of a loop, and put the caller's settings back afterwards (an embedding application keeps its
    own pools). Why it matters here: the per-clip host work of the detector (score filter, tracker assignment) is a handful of tiny ops,
    but a pool sized to the 256 logical CPUs a container SEES wakes 256 spinning workers for each of them; under a 16-CPU cgroup quota
    that exhausts the CFS budget and the whole process is throttled for the rest of the 100 ms period - measured as random 10 - 80 ms
    stalls per clip (round 4; /sys/fs/cgroup/cpu.stat: nr_throttled). `info` says what was done."""

    def __init__(self, local_rank=0, world=1, pin=None):
        self.args = (local_rank, world, pin)
        self.info = None

    def __enter__(self):
        self._threads = torch.get_num_threads()
        try:
            self._affinity = os.sched_getaffinity(0)
        except (AttributeError, OSError):
            self._affinity = None
        self.info, self._blas = _size_host_pools(*self.args)
        return self

    def __exit__(self, *exc):
        torch.set_num_threads(self._threads)
        if self._blas is not None:
            try:
                self._blas.restore_original_limits()
            except Exception:
                pass
        if self._affinity is not None and self.info.get("cpus") is not None:
            try:
                os.sched_setaffinity(0, self._affinity)
            except OSError:
                pass
        return False


_blas_limit = None


def size_host_pools(local_rank=0, world=1, pin=None):
    """The same sizing for the REST OF THE PROCESS (a rank of bench.py: nothing else lives in it). Library entry points use the
    `host_pools` context manager instead, which restores the caller's settings. Returns what was done."""
    global _blas_limit
    info, _blas_limit = _size_host_pools(local_rank, world, pin)       # kept alive: the BLAS limit lasts as long as the object
    return info


def _size_host_pools(local_rank=0, world=1, pin=None):
    info = {"threads": torch.get_num_threads(), "cpus": None}
    blas = None
    try:
        allowed = sorted(os.sched_getaffinity(0))
        n = max(1, min(len(allowed), host_cores()) // max(1, world))
        mine = allowed[local_rank * n:(local_rank + 1) * n] or allowed
        if world > 1 if pin is None else pin:
            os.sched_setaffinity(0, mine)
            info["cpus"] = [mine[0], mine[-1]]
        torch.set_num_threads(max(1, min(torch.get_num_threads(), n)))
        try:
            from threadpoolctl import threadpool_limits
            blas = threadpool_limits(limits=max(1, min(n, 16)))
        except Exception:                                                      # threadpoolctl is optional
            pass
        info["threads"] = torch.get_num_threads()
        info["host_cores"] = n
    except (AttributeError, OSError) as e:
        info["error"] = f"{type(e).__name__}: {e}"[:80]
    return info, blas


def dist_env():
    """(rank, local_rank, world_size) from the torch.distributed.run environment (defaults: single process)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_distributed(backend=None, single_rank_group=False):
    """Initialise torch.distributed when launched with WORLD_SIZE > 1. backend: 'nccl' (= RCCL on ROCm)
    on GPUs, 'gloo' on CPU. single_rank_group: create the process group for a world of one as well (the collectives then run
    through the backend with one member: how the RCCL path is exercised on a one-GPU box). Returns (rank, local_rank, world)."""
    rank, local_rank, world = dist_env()
    if (world > 1 or single_rank_group) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if backend == "nccl":                            # (gloo ranks make no HIP call here: `bench.py --dry-run-cpu` must not touch the GPU)
            if local_rank >= torch.cuda.device_count():
                raise RuntimeError(f"rank {rank}: local rank {local_rank} has no GPU ({torch.cuda.device_count()} visible)")
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


# ---------------------------------------------------------------------------------------------------
# Per-clip result gather, overlapped with the next clip
# ---------------------------------------------------------------------------------------------------
def clip_result_template(T, H, W, device, max_segments=100):
    """Tensors a rank ships per clip: the harness' per-frame outputs as it stores them (tools/test_vpq.py:44-46 casts
    `panoptic_outputs` / `fcn_outputs` to uint8) plus up to `max_segments` (class, probability, object id) triples per
    frame and their count."""
    return {"panoptic_outputs": torch.zeros((T, H, W), dtype=torch.uint8, device=device),
            "fcn_outputs": torch.zeros((T, H, W), dtype=torch.uint8, device=device),
            "segments": torch.zeros((T, max_segments, 3), dtype=torch.float32, device=device),
            "num_segments": torch.zeros((T,), dtype=torch.int32, device=device)}


def pack_clip_result(frame_dicts, template):
    """Result dicts of `VPS_Temporal_Slots.clip_test` (one per frame) -> the fixed-shape tensors of `clip_result_template`."""
    out = {k: torch.zeros_like(v) for k, v in template.items()}
    for t, r in enumerate(frame_dicts):
        out["panoptic_outputs"][t].copy_(r["panoptic_outputs"][0].to(torch.uint8))
        out["fcn_outputs"][t].copy_(r["fcn_outputs"][0].to(torch.uint8))
        n = min(len(r["panoptic_cls_inds"]), out["segments"].shape[1])
        out["num_segments"][t] = n
        if n:
            seg = torch.stack([torch.as_tensor(r["panoptic_cls_inds"][:n]).float(), torch.as_tensor(r["panoptic_cls_prob"][:n]).float(),
                               torch.as_tensor(r["panoptic_det_obj_ids"][:n]).float()], dim=1)
            out["segments"][t, :n].copy_(seg)
    return out


class ClipResultGatherer:
    """Gathers a fixed set of result tensors to rank 0 once per clip (or per step), WITHOUT stopping the compute stream:
    `submit` copies the results into one of `depth` staging sets on the caller's stream, then issues the gathers
    asynchronously behind an event on a side stream (RCCL runs them on its own stream; rank 0 receives from every peer
    over that peer's direct xGMI link); the caller goes on with the next clip and a staging set is only waited for when
    it comes round again. `drain` waits for everything in flight. No collective touches the data path of the kernels.

    No process group (a plain single process): the staging copy is the whole operation. With a process group the gathers are
    issued whatever its size (a group of one still goes through the backend)."""

    def __init__(self, template, depth=2):
        self.collective = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size() if self.collective else 1
        self.rank = dist.get_rank() if self.collective else 0
        self.names = sorted(template)
        self.depth = max(1, depth)
        self.device = next(iter(template.values())).device
        self.stage = [{k: torch.empty_like(template[k]) for k in self.names} for _ in range(self.depth)]
        self.recv = None
        if self.collective and self.rank == 0:
            self.recv = [{k: [torch.empty_like(template[k]) for _ in range(self.world)] for k in self.names}
                         for _ in range(self.depth)]
        self.pending = [[] for _ in range(self.depth)]
        self.n = 0
        self.cuda = self.device.type == "cuda"
        self.side = torch.cuda.Stream(device=self.device) if self.cuda else None
        self.bytes_per_submit = sum(t.numel() * t.element_size() for t in template.values())

    def _wait(self, d):
        for w in self.pending[d]:
            w.wait()
        self.pending[d] = []

    def submit(self, tensors):
        """tensors: name -> tensor shaped like the template. Returns the staging index used."""
        d = self.n % self.depth
        self.n += 1
        self._wait(d)                                     # this staging set was handed to a gather `depth` submits ago
        st = self.stage[d]
        for k in self.names:
            st[k].copy_(tensors[k], non_blocking=True)
        if not self.collective:
            return d
        if self.cuda:
            self.side.wait_stream(torch.cuda.current_stream(self.device))
            ctx = torch.cuda.stream(self.side)
        else:
            import contextlib
            ctx = contextlib.nullcontext()
        with ctx:
            for k in self.names:
                gl = self.recv[d][k] if self.rank == 0 else None
                self.pending[d].append(dist.gather(st[k], gather_list=gl, dst=0, async_op=True))
        return d

    def drain(self):
        for d in range(self.depth):
            self._wait(d)
        if self.cuda:
            torch.cuda.current_stream(self.device).wait_stream(self.side)

    def last(self, d):
        """Rank 0, after the staging set `d` has been waited for: name -> list over ranks of the gathered tensors."""
        if not self.collective:
            return {k: [self.stage[d][k]] for k in self.names}
        return self.recv[d] if self.rank == 0 else None
