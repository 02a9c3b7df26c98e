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


def _run(cmd, env=None, only_line=True, timeout=300):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)
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
    # on-chip kernel: VALU-issue roofline from the build's own assembly x the iterations really run, 0 < frac <= 1
    r = d["roofline"]
    assert r["bound"] == "valu" and r["peak"] == 1228.8 and r["launches"] == 2 and "fused_split_kernel<float, 1" in r["kernel"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.2 < r["frac"] <= 1.0 and 0.2 < r["valu_pipe_busy_frac"] <= 1.0
    assert r["frame_turns_timed"] == 2 * 2048 * 50 and 900 < r["valu_instr_per_wave_turn"] < 1100 and r["waves_per_frame"] == 4
    # HBM bytes per launch: measured in this very run (child runs under rocprofv3 --pmc) where the profiler is there, else the committed pass
    assert r["traffic"] is None or (r["traffic"] > 0 and ("committed" in r["traffic_source"] or "measured in this run" in r["traffic_source"]))
    import shutil
    if shutil.which("rocprofv3"):
        assert "measured in this run" in r["traffic_source"], r["traffic_source"]
        # on-chip state: the LLRs in (4 bytes per bit), a byte per bit out, byte-granular writes -- between one and three times that
        assert 2048 * 5632 * 5 <= r["traffic"] <= 3 * 2048 * 5632 * 5, r["traffic"]
        # ... and how busy the vector pipes were, by the shader counters of a third child run (2 048 frames: 8 workgroups per CU's worth of work)
        assert 0.1 < r["counters"]["valu_pipe_busy"] < 1.5 and 0.0 < r["counters"]["waves_waiting"] < 1.0, r["counters"]
    # the contract's HBM byte model rides along, priced with the same iteration sum
    h = d["roofline_hbm_model"]
    assert h["bound"] == "hbm" and h["unit"] == "GB/s" and h["peak"] == 8000.0 and h["mean_iters_timed"] == 50.0
    # the timed kernel against the flood path, below and inside the waterfall
    w = d["proof_of_work"]
    assert w["ok"] and len(w["points"]) == 2 and w["points"][1]["converged_frac"] > 0.5 and w["points"][1]["distinct_iteration_counts"] > 3
    assert w["points"][1]["decoded_bit_errors"] < w["points"][1]["channel_bit_errors"] / 10
    assert d["metric"] == "decoded info Mbit/s @ 50 BP iters, jpl.4096.4.5, Eb/N0=2 dB"
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "Mbit/s" and c["cores"] >= 1 and c["value"] > 0 and "frames" in c["sample"]
    assert d["value"] > 100 * c["value"]
    # value = frames * k / time
    assert abs(d["value"] - 2 * 2048 * 4096 / (d["ms_per_step"] * 2e-3) / 1e6) / d["value"] < 0.02
    # the other BASELINE configurations ride along as labelled measurements (never as `value`)
    assert "fused_pk16_kernel" in d["fp16_packed"]["kernel"] and "layered_lds_kernel" in d["long_code_layered"]["kernel"]
    assert d["configs0_moon_tanh"]["code_name"] == "ldpc/hip-tanh/moon.7.13/20" and len(d["configs0_moon_tanh"]["points"]) == 2
    assert "fused_split_kernel" in d["configs1_jpl1024_minsum"]["kernel"] and d["configs1_jpl1024_minsum"]["points"][0]["value"] > 0
    sweep = d["configs2_mackay_tanh_sweep"]
    assert [p_["ebn0_db"] for p_ in sweep["points"]] == [1.0, 2.0, 3.0, 4.0] and sweep["kernel"].endswith(", true>")
    assert sweep["points"][0]["mean_iters"] > sweep["points"][3]["mean_iters"] and sweep["points"][3]["ber"] < sweep["points"][0]["ber"]


def test_two_ranks_rehearsal_aggregates_over_ranks():
    d = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
              "--master-port", "29533", "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2048", "--cpu-seconds", "0"],
             env={"LDPC_BENCH_REHEARSE": "1"}, only_line=False)
    assert d["n_gpus"] == 2 and "rehearsal" in d and "cpu_baseline" not in d
    assert abs(d["value"] - 2 * 2 * 2048 * 4096 / (d["ms_per_step"] * 2e-3) / 1e6) / d["value"] < 0.02   # whole-job aggregate


def test_bench_starts_its_own_ranks():
    """The driver's command shape for N > 1 WITHOUT a launcher: `python3 bench.py --gpus 2 --steps 2 --warmup 1` (the headline
    workload at its full batch).  The parent never touches the GPU; two child ranks share this box's one GPU (rehearsal:
    gloo instead of RCCL, which refuses two ranks on one device).  One JSON line, both ranks took part in the collectives."""
    env = {"LDPC_BENCH_REHEARSE": "1"}
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        os.environ.pop(k, None)
    d = _run(["python3", "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1"], env=env, only_line=False, timeout=900)
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and len(d["per_rank_ms_per_step"]) == 2 and d["launcher"] == "self"
    assert d["steps"] == 2 and d["warmup"] == 1 and d["config"]["batch_per_gpu"] == 65536 and "cpu_baseline" not in d
    assert max(d["per_rank_ms_per_step"]) == d["ms_per_step"]
    assert abs(d["value"] - 2 * 2 * 65536 * 4096 / (d["ms_per_step"] * 2e-3) / 1e6) / d["value"] < 0.02   # whole-job aggregate
    assert d["proof_of_work"]["ok"]


def test_bench_starts_its_own_ranks_on_a_fresh_jit_cache(tmp_path):
    """Two ranks that both need a run-time specialised kernel nobody has built yet (empty cache directory): they compile
    it concurrently, each renames its own temp file into place, both load a whole code object."""
    from tests.helpers import synthetic   # a synthetic QC shape goes through jit.cc; bench.py takes shipped codes only, so drive the library directly
    import textwrap
    child = tmp_path / "rank.py"
    child.write_text(textwrap.dedent(f"""
        import os, sys
        sys.path.insert(0, {ROOT!r})
        import numpy as np
        import ecc_ldpc_amd as E
        from tests.helpers import synthetic
        from oracle import oracle
        E.init(0)
        c = synthetic("small-2x4-sz32")
        _, llr = c.frames(32, 4.0, seed=5)
        dec = E.Decoder(c.hip_code(E), "min", "f32", 32)
        assert dec.kernel_name.startswith("ldpc_jit_split_")
        bits, its, conv = dec.decode_batch(llr.astype(np.float32), 30)
        ob, oi, oc = oracle.decode_batch(c.graph, "min", 30, llr, nthreads=2)
        assert np.array_equal(bits, ob) and np.array_equal(conv, oc)
        if os.environ["RANK"] == "0":
            print("ok", sorted(os.listdir(os.environ["LDPC_JIT_CACHE"])))
    """))
    import io
    from ecc_ldpc_amd import launch
    cache = tmp_path / "cache"
    cache.mkdir()
    os.environ["LDPC_JIT_CACHE"] = str(cache)
    try:
        out, err = io.BytesIO(), io.BytesIO()
        rc = launch.launch_ranks(2, [sys.executable, str(child)], timeout=600, stdout=out, stderr=err)
    finally:
        del os.environ["LDPC_JIT_CACHE"]
    assert rc == 0, err.getvalue().decode()[-2000:]
    files = os.listdir(cache)
    assert out.getvalue().decode().startswith("ok") and len(files) == 1 and files[0].endswith(".hsaco")


def test_early_exit_is_priced_by_the_iterations_run():
    """4 dB: frames converge after a few turns; both roofline objects must price what ran, not max_iters."""
    d = _run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--batch", "2048", "--cpu-seconds", "0", "--ebn0", "4"])
    r, h = d["roofline"], d["roofline_hbm_model"]
    assert d["metric"].endswith("Eb/N0=4 dB") and d["mean_iters"] < 10
    assert r["frame_turns_timed"] < 2 * 2048 * 10 and 0 < r["frac"] <= 1.0
    assert abs(h["mean_iters_timed"] - d["mean_iters"]) < 1.0


def test_flood_path_reports_an_hbm_roofline():
    d = _run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--batch", "2048", "--cpu-seconds", "0", "--path", "flood"])
    r = d["roofline"]
    # QC code: the frame-per-workgroup flooding kernel (one launch per batch); LDPC_FLOOD_QC=0 gives the batch-major pair
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and "flood_qc_kernel" in r["kernel"] and 0 < r["frac"] <= 1.0 and d["proof_of_work"]["ok"]
    d2 = _run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--batch", "2048", "--cpu-seconds", "0", "--path", "flood"],
              env={"LDPC_FLOOD_QC": "0"})
    assert "flood_cn" in d2["roofline"]["kernel"] and 0 < d2["roofline"]["frac"] <= 1.0 and d2["proof_of_work"]["ok"]


def test_rccl_collectives_run_with_one_rank():
    """The N > 1 path's process-group setup and its two collectives (all-reduce of the tallies, max of the elapsed time) over
    the real RCCL backend, with a single rank on this 1-GPU box: next to the library's own HIP client in the process."""
    d = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
              "--master-port", "29534", "bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "2048", "--cpu-seconds", "0"],
             env={"LDPC_BENCH_FORCE_DIST": "1"}, only_line=False)
    assert d["n_gpus"] == 1 and "rehearsal" not in d and d["proof_of_work"]["ok"]
    assert abs(d["value"] - 2 * 2048 * 4096 / (d["ms_per_step"] * 2e-3) / 1e6) / d["value"] < 0.02
