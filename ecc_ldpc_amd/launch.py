"""One process per GPU, started from a parent that never touches the GPU.

`python3 bench.py --gpus N` (N > 1, no WORLD_SIZE in the environment) lands here BEFORE torch or the HIP library is
imported: the parent starts N children of the same command line -- rank r gets RANK / LOCAL_RANK = r, WORLD_SIZE /
LOCAL_WORLD_SIZE = N, MASTER_ADDR = 127.0.0.1 and a free MASTER_PORT, the variables `torch.distributed.run` would
set -- relays rank 0's stdout (the ONE JSON line) to its own stdout, the other ranks' stdout and every stderr to its
stderr, and returns the worst exit code.  Nothing is exec'ed from a process that has initialised the GPU: the
children are fresh interpreters, the parent only waits.  If a rank dies the others are sent SIGTERM (exact PIDs)
after a short grace period, so a failed collective cannot leave ranks blocked in a rendezvous.

(The reference's counterpart is `maxThreadCount` decoder replicas inside one process, src/ECC/Code/LDPC/Utils.hs:53,
63-69; the one-process form of this framework is `ecc-ldpc-hip -d0,1,...`.)

Standard library only: importing this module loads neither torch nor libldpc_hip.so.
"""
from __future__ import annotations

import os
import signal
import socket
import subprocess
import sys
import threading
import time


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_env(rank: int, world: int, port: int, base=None) -> dict:
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LDPC_BENCH_LAUNCHER="self")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this pool
    env.setdefault("OMP_NUM_THREADS", "1")
    return env


def _relay(src, dst, prefix=b""):
    for line in iter(src.readline, b""):
        try:
            dst.write(prefix + line)
            dst.flush()
        except (BrokenPipeError, ValueError):
            break
    src.close()


def launch_ranks(world: int, argv: list, timeout: float = 1500.0, grace: float = 10.0, stdout=None, stderr=None) -> int:
    """Start `argv` once per rank, wait for all of them, return the worst exit code (a rank killed by signal s counts
    as 128 + s; a timeout as 124)."""
    stdout = stdout or sys.stdout.buffer
    stderr = stderr or sys.stderr.buffer
    port = free_port()
    procs, threads = [], []
    for r in range(world):
        p = subprocess.Popen(argv, env=rank_env(r, world, port), stdout=subprocess.PIPE, stderr=subprocess.PIPE, stdin=subprocess.DEVNULL)
        procs.append(p)
        tag = f"[rank {r}] ".encode()
        for src, dst, pre in ((p.stdout, stdout if r == 0 else stderr, b"" if r == 0 else tag), (p.stderr, stderr, tag)):
            t = threading.Thread(target=_relay, args=(src, dst, pre), daemon=True)
            t.start()
            threads.append(t)
    deadline = time.monotonic() + timeout
    failed_at = None
    worst = 0
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        now = time.monotonic()
        if failed_at is None and any(c not in (None, 0) for c in codes):
            failed_at = now
        if now > deadline:
            worst = 124
            failed_at = failed_at or now - grace
        if failed_at is not None and now - failed_at >= grace:
            for p in procs:          # exactly the processes started above
                if p.poll() is None:
                    p.send_signal(signal.SIGTERM)
            t_kill = time.monotonic() + grace
            while time.monotonic() < t_kill and any(p.poll() is None for p in procs):
                time.sleep(0.05)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            for p in procs:
                p.wait()
            break
        time.sleep(0.05)
    for t in threads:
        t.join(timeout=5)
    for p in procs:
        c = p.returncode
        c = 128 - c if c is not None and c < 0 else (c or 0)
        worst = max(worst, c)
    return worst
