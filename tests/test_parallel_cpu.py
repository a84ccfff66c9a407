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


def _gather_worker(rank, world, port, q, T=3, H=16, W=32):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from slotvps_amd import parallel
    parallel.init_distributed(backend="gloo")
    tmpl = parallel.clip_result_template(T, H, W, torch.device("cpu"), max_segments=100)
    gat = parallel.ClipResultGatherer(tmpl, depth=2)
    last = None
    for clip in range(5):                               # five clips per rank, two staging sets: sets are reused
        frames = []
        for t in range(T):
            k = (rank + clip + t) % 4
            frames.append({"panoptic_outputs": torch.full((1, H, W), 10 * rank + clip, dtype=torch.int64),
                           "fcn_outputs": torch.full((1, H, W), t, dtype=torch.int64),
                           "panoptic_cls_inds": torch.arange(1, k + 1), "panoptic_cls_prob": torch.full((k,), 0.9),
                           "panoptic_det_obj_ids": torch.arange(k) + 100 * rank})
        last = gat.submit(parallel.pack_clip_result(frames, tmpl))
    gat.drain()
    if rank == 0:
        got = gat.last(last)
        q.put({"pan": [int(got["panoptic_outputs"][r][0, 0, 0]) for r in range(world)],
               "fcn": [int(got["fcn_outputs"][r][2, 0, 0]) for r in range(world)],
               "nseg": [got["num_segments"][r].tolist() for r in range(world)],
               "ids": [got["segments"][r][2, :, 2].tolist()[:3] for r in range(world)],
               "uniform": [bool((got["panoptic_outputs"][r] == got["panoptic_outputs"][r][0, 0, 0]).all()) for r in range(world)],
               "ranks_seen": len(got["panoptic_outputs"]), "bytes": gat.bytes_per_submit})
    parallel.barrier()
    torch.distributed.destroy_process_group()


def test_per_clip_gather_overlapped_two_ranks():
    """SURVEY 8e payload (uint8 panoptic + semantic maps [T, H, W] and <= 100 (class, prob, id) triples per frame),
    gathered once per clip through the double-buffered asynchronous gatherer."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert out["pan"] == [4, 14] and out["fcn"] == [2, 2]                  # the LAST clip (index 4) of each rank
    assert out["nseg"] == [[(0 + 4 + t) % 4 for t in range(3)], [(1 + 4 + t) % 4 for t in range(3)]]
    assert out["ids"][1][:3] == [100.0, 101.0, 102.0]                       # rank 1, frame 2: k = (1 + 4 + 2) % 4 = 3 segments
    assert out["bytes"] == 3 * 16 * 32 * 2 + 3 * 100 * 3 * 4 + 3 * 4


def test_per_clip_gather_eight_ranks_full_size_payload():
    """The N = 8 job of BASELINE config 3 on CPU ranks: eight gloo ranks, five clips each through the double-buffered gatherer with
    the SURVEY 8e payload at its real size (T = 5, 1024 x 2048: 2 x 10.5 MB of uint8 maps + the segment triples per clip and rank);
    rank 0 must hold the LAST clip of every one of the eight ranks, intact."""
    world, T, H, W = 8, 5, 1024, 2048
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, q, T, H, W)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=600)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert out["ranks_seen"] == world
    assert out["pan"] == [10 * r + 4 for r in range(world)] and out["fcn"] == [2] * world and all(out["uniform"])
    assert out["nseg"] == [[(r + 4 + t) % 4 for t in range(T)] for r in range(world)]
    assert out["bytes"] == T * H * W * 2 + T * 100 * 3 * 4 + T * 4


def test_bench_launcher_eight_ranks_dry_run():
    """`python bench.py --gpus 8 --dry-run-cpu`: the launcher, rendezvous, per-step gather with per-rank checksums and rank 0's line for
    the N = 8 job (gloo, no kernels): n_gpus = 8, a payload from each of the eight ranks."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1", "--clips-per-launch", "2",
                        "--dry-run-cpu"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["world_size"] == 8 and line["gather_ok"] is True and line["payload_ranks"] == 8
    assert len(line["per_rank_ms_per_step"]) == 8


def test_bench_launcher_starts_the_ranks_itself():
    """`python bench.py --gpus 2` (no torch.distributed environment) must start two ranks, report n_gpus = 2 from rank 0 and
    exit 0; a WORLD_SIZE that contradicts --gpus must be refused. Rehearsed with --dry-run-cpu (gloo, no kernels)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                        "--dry-run-cpu"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["world_size"] == 2 and line["gather_ok"] is True and line["steps"] == 4
    env["WORLD_SIZE"] = "3"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run-cpu"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "refusing" in r.stderr


def _one_rank_worker(port, q):
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from slotvps_amd import parallel
    parallel.init_distributed(backend="gloo", single_rank_group=True)
    tmpl = parallel.clip_result_template(2, 8, 16, torch.device("cpu"))
    gat = parallel.ClipResultGatherer(tmpl, depth=2)
    assert gat.collective and gat.world == 1
    for i in range(3):
        d = gat.submit({k: torch.full_like(v, i + 1) for k, v in tmpl.items()})
    gat.drain()
    q.put((d, int(gat.last(d)["fcn_outputs"][0][0, 0, 0]), parallel.max_over_ranks(2.5, torch.device("cpu"))))
    torch.distributed.destroy_process_group()


def test_group_of_one_goes_through_the_backend():
    """A process group of ONE rank still issues the gathers (the CPU rehearsal of tests/test_parallel_gpu.py::test_one_rank_nccl_group)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_one_rank_worker, args=(_free_port(), q))
    p.start()
    out = q.get(timeout=120)
    p.join(timeout=120)
    assert p.exitcode == 0 and out == (0, 3, 2.5)
