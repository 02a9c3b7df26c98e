"""GPU: bench.py's output contract -- ONE JSON line with the driver's keys, the roofline and cpu_baseline objects,
for the headline workload at a small batch and for two ranks sharing the GPU in rehearsal mode."""
import json
import os
import subprocess
import sys

import pytest

from tests.helpers import ROOT

pytestmark = pytest.mark.gpu

KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline"}


def _run(cmd, env=None, only_line=True):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    if only_line:
        assert len(lines) == 1, p.stdout      # exactly one line on stdout
    js = [l for l in lines if l.startswith("{")]
    assert len(js) == 1, p.stdout             # (the gloo transport of the rehearsal prints a banner of its own)
    return json.loads(js[0])


def test_one_json_line_with_roofline_and_cpu_baseline():
    d = _run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--batch", "2048", "--cpu-seconds", "1"])
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["unit"] == "Mbit/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["launches"] == 2
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["kernel"] == "fused_split_kernel"
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "Mbit/s" and c["cores"] >= 1 and c["value"] > 0 and "frames" in c["sample"]
    assert d["value"] > 100 * c["value"]
    # value = frames * k / time
    assert abs(d["value"] - 2 * 2048 * 4096 / (d["ms_per_step"] * 2e-3) / 1e6) / d["value"] < 0.02


def test_two_ranks_rehearsal_aggregates_over_ranks():
    d = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
              "--master-port", "29533", "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2048", "--cpu-seconds", "0"],
             env={"LDPC_BENCH_REHEARSE": "1"}, only_line=False)
    assert d["n_gpus"] == 2 and "rehearsal" in d and "cpu_baseline" not in d
    assert abs(d["value"] - 2 * 2 * 2048 * 4096 / (d["ms_per_step"] * 2e-3) / 1e6) / d["value"] < 0.02   # whole-job aggregate
