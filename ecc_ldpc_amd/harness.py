"""Multi-GPU sharding of a BER / throughput run: the part of the harness that is independent of the
device.  Frames are independent, so rank r of P decodes global frame ids
    [first_frame + r * per_rank, first_frame + (r + 1) * per_rank)
generated locally from the counter-based RNG (no scatter), and the ONLY exchange is one all-reduce
(sum) of four int64 tallies {frames, frame errors, bit errors, iterations} -- RCCL over xGMI when the
backend is "nccl", gloo in the CPU tests.  (The reference has no distributed path at all:
SURVEY.md section 2 rows 24-25.)"""
from __future__ import annotations

from dataclasses import dataclass


@dataclass
class Shard:
    rank: int
    world: int
    first_frame: int
    frames: int


def shard_frames(total_frames: int, rank: int, world: int, first_frame: int = 0) -> Shard:
    """Contiguous block partition; the first (total % world) ranks take one extra frame."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, extra = divmod(int(total_frames), world)
    mine = base + (1 if rank < extra else 0)
    start = first_frame + rank * base + min(rank, extra)
    return Shard(rank, world, start, mine)


def all_reduce_tallies(tally, dist=None):
    """tally: torch int64 tensor [4] on the rank's device -> summed over ranks in place."""
    if dist is not None and dist.is_initialized():      # (also with one rank: the collective itself is exercised)
        if dist.get_backend() == "gloo" and tally.is_cuda:  # rehearsal of the N > 1 path without RCCL
            host = tally.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            tally.copy_(host)
        else:
            dist.all_reduce(tally, op=dist.ReduceOp.SUM)
    return tally


def max_over_ranks(seconds: float, device, dist=None) -> float:
    import torch
    multi = dist is not None and dist.is_initialized()
    if multi and dist.get_backend() == "gloo":
        device = "cpu"
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if multi:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def per_rank(value: float, device, dist=None) -> list:
    """-> [value of rank 0, value of rank 1, ...] on every rank (one all-gather of a double).  Its length is the number
    of ranks that really took part in the collective: bench.py prints it as `ranks_seen`."""
    import torch
    multi = dist is not None and dist.is_initialized()
    if not multi:
        return [float(value)]
    if dist.get_backend() == "gloo":
        device = "cpu"
    mine = torch.tensor([value], dtype=torch.float64, device=device)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [float(t.item()) for t in out]


def summarize(tally, k: int):
    frames, ferr, berr, its = [int(v) for v in tally.tolist()]
    return {"frames": frames, "fer": ferr / max(frames, 1), "ber": berr / max(frames * k, 1), "mean_iters": its / max(frames, 1)}
