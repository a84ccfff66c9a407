"""Device time per call through a hipGraph (shared by the slot-side probes)."""
import torch


def timed_graph(fn, n=20):
    """n calls captured into one hipGraph (no host launch overhead between them), replayed 5 times -> (us per call, last result)."""
    for _ in range(3):
        o = fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(n):
                o = fn()
    torch.cuda.current_stream().wait_stream(side)
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3, o
