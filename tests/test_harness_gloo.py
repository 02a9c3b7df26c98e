"""CPU, world_size 2 over gloo: the N > 1 logic of bench.py (frame sharding, tally all-reduce,
max-over-ranks timing) without a GPU."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ecc_ldpc_amd import harness


def test_shard_partition_covers_every_frame_once():
    for total in (0, 1, 7, 64, 1001):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                s = harness.shard_frames(total, r, world, first_frame=100)
                seen += list(range(s.first_frame, s.first_frame + s.frames))
            assert seen == list(range(100, 100 + total))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = harness.shard_frames(1001, rank, world)
    # fake per-rank results: every 10th global frame is in error with 3 bit errors, 50 iterations each
    ids = torch.arange(s.first_frame, s.first_frame + s.frames)
    bad = (ids % 10 == 0)
    tally = torch.tensor([s.frames, int(bad.sum()), int(bad.sum()) * 3, s.frames * 50], dtype=torch.int64)
    harness.all_reduce_tallies(tally, dist)
    t = harness.max_over_ranks(0.5 + rank, torch.device("cpu"), dist)
    q.put((rank, tally.tolist(), t))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_tally_allreduce_and_timing():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, tally, t in res:
        assert tally == [1001, 101, 303, 1001 * 50]  # identical on both ranks: whole-job totals
        assert t == 1.5                               # max over ranks
    s = harness.summarize(torch.tensor(res[0][1]), k=4096)
    assert s["frames"] == 1001 and abs(s["fer"] - 101 / 1001) < 1e-12
