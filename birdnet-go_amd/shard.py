"""Multi-GPU layout of the hot path: one process per GPU, clips shard embarrassingly.

The reference runs one inference at a time (`internal/classifier/orchestrator.go:531` inferenceMu),
so there is no reference counterpart; BASELINE.json's config 3 defines the layout: index-contiguous
clip shards per rank, ONE collective — a broadcast of the frozen model bytes from rank 0 (RCCL over
xGMI on GPUs; gloo in the CPU tests) — and no per-batch exchange.
"""
import numpy as np


def shard_range(n_items: int, rank: int, world: int):
    """Index-contiguous shard [lo, hi) of n_items for `rank` (remainder spread over the first ranks)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def broadcast_model_bytes(blob, src=0, device=None):
    """Broadcast the model blob from `src` to every rank with torch.distributed (backend nccl == RCCL
    on ROCm; gloo on CPU).  Non-src ranks pass blob=None.  Returns bytes on every rank."""
    import torch
    import torch.distributed as dist

    rank = dist.get_rank()
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    n = torch.zeros(1, dtype=torch.int64, device=dev)
    if rank == src:
        n[0] = len(blob)
    dist.broadcast(n, src=src)
    size = int(n.item())
    if rank == src:
        t = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
    else:
        t = torch.empty(size, dtype=torch.uint8, device=dev)
    dist.broadcast(t, src=src)
    return t.cpu().numpy().tobytes()


def gather_results(local: np.ndarray, world: int):
    """Test helper: all-gather per-rank result arrays (not used on the timed path)."""
    import torch
    import torch.distributed as dist

    objs = [None] * world
    dist.all_gather_object(objs, local)
    return objs
