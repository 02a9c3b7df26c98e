"""CPU: the self-launcher behind `python3 bench.py --gpus N` (ecc_ldpc_amd/launch.py) -- one child per rank with the
rendezvous variables torch.distributed.run would set, rank 0's stdout relayed alone, worst exit code returned, a dead
rank takes the others down instead of leaving them in a rendezvous.  And bench.py takes that route before it imports
torch or the HIP library."""
import io
import json
import os
import subprocess
import sys
import textwrap
import time

from ecc_ldpc_amd import launch
from tests.helpers import ROOT


def _child(tmp_path, body):
    p = tmp_path / "child.py"
    p.write_text(textwrap.dedent(body))
    return [sys.executable, str(p)]


def test_rank_environment_and_single_line_relay(tmp_path):
    argv = _child(tmp_path, """
        import json, os, sys
        r = int(os.environ["RANK"])
        print(json.dumps({k: os.environ[k] for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LDPC_BENCH_LAUNCHER")}))
        print("note from rank", r, file=sys.stderr)
    """)
    out, err = io.BytesIO(), io.BytesIO()
    assert launch.launch_ranks(3, argv, stdout=out, stderr=err) == 0
    lines = out.getvalue().decode().splitlines()
    assert len(lines) == 1                       # rank 0's line only
    d = json.loads(lines[0])
    assert d["RANK"] == "0" and d["WORLD_SIZE"] == "3" and d["LOCAL_WORLD_SIZE"] == "3" and d["MASTER_ADDR"] == "127.0.0.1" and d["LDPC_BENCH_LAUNCHER"] == "self"
    e = err.getvalue().decode()
    others = [json.loads(l.split("] ", 1)[1]) for l in e.splitlines() if l.startswith("[rank ") and "{" in l]
    assert sorted(o["RANK"] for o in others) == ["1", "2"] and all(o["MASTER_PORT"] == d["MASTER_PORT"] for o in others)
    assert all(f"[rank {r}] note from rank {r}" in e for r in range(3))


def test_worst_exit_code_and_dead_rank_stops_the_others(tmp_path):
    argv = _child(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(600)     # a rank blocked in a rendezvous its peer never reaches
    """)
    t0 = time.monotonic()
    rc = launch.launch_ranks(2, argv, grace=0.5, stdout=io.BytesIO(), stderr=io.BytesIO())
    assert time.monotonic() - t0 < 30
    assert rc == 128 + 15                        # rank 0 ended by SIGTERM (143) outranks rank 1's 7


def test_timeout_is_reported_as_124_or_worse(tmp_path):
    argv = _child(tmp_path, "import time; time.sleep(600)")
    rc = launch.launch_ranks(2, argv, timeout=0.5, grace=0.3, stdout=io.BytesIO(), stderr=io.BytesIO())
    assert rc >= 124


def test_bench_py_self_launches_before_importing_torch(tmp_path):
    """`python3 bench.py --gpus 2` with no WORLD_SIZE: the parent must not import torch or load the HIP library; here (no
    GPU) each child fails in ldpc_init with ENODEVICE and the parent returns non-zero with both ranks' messages."""
    probe = tmp_path / "sitecustomize.py"     # records which modules the PARENT has loaded when it exits
    probe.write_text(textwrap.dedent("""
        import atexit, os, sys
        if "RANK" not in os.environ and os.environ.get("LDPC_PROBE_OUT"):
            def _dump():
                open(os.environ["LDPC_PROBE_OUT"], "w").write("\\n".join(sorted(sys.modules)))
            atexit.register(_dump)
    """))
    env = dict(os.environ, PYTHONPATH=str(tmp_path) + os.pathsep + os.environ.get("PYTHONPATH", ""), LDPC_PROBE_OUT=str(tmp_path / "mods.txt"))
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "64", "--cpu-seconds", "0"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    mods = (tmp_path / "mods.txt").read_text().split()
    assert "torch" not in mods and "ecc_ldpc_amd.launch" in mods
    import torch
    if not torch.cuda.is_available():
        assert p.returncode != 0 and "[rank 0]" in p.stderr and "[rank 1]" in p.stderr
