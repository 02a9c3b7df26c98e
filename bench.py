#!/usr/bin/env python3
"""bench.py -- decoded information Mbit/s of the LDPC BP decode path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = one pass of the hot path (ldpc_decode_batch_dev) over one batch of synthetic AWGN frames
that are already resident in HBM.  Workload at every N: BASELINE.json's metric configuration --
codes/jpl.4096.4.5, rate 4/5 puncturing, min-sum, 50 iterations, Eb/N0 = 2 dB -- with a fixed
per-GPU batch (weak scaling: frames are independent, each rank generates its own frame-id range
from the counter-based RNG; the only exchange is one RCCL all-reduce of the error tallies).
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--code", default="jpl.4096.4.5")
    ap.add_argument("--rate", default="4/5")
    ap.add_argument("--variant", default="minsum", choices=["minsum", "tanh"])
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64", "f16"])
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--ebn0", type=float, default=2.0)
    ap.add_argument("--batch", type=int, default=65536, help="frames per GPU per step")
    ap.add_argument("--path", default="auto", choices=["auto", "flood", "fused"])
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the cpu_baseline sample (0 = skip)")
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5EEDC0DE)
    args = ap.parse_args()

    import numpy as np
    import torch
    import ecc_ldpc_amd as E

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus > 1 launch through torch.distributed.run (one process per GPU)")
    # LDPC_BENCH_REHEARSE=1: run the N > 1 code path on a box with fewer GPUs than ranks (ranks share devices,
    # gloo instead of RCCL).  For checking the launch/sharding/tally logic only; its number is not a result.
    rehearse = os.environ.get("LDPC_BENCH_REHEARSE") == "1"
    ndev = torch.cuda.device_count()
    # (a launcher that gives every rank its own HIP_VISIBLE_DEVICES shows each rank ONE device: index 0)
    dev_index = local_rank % ndev if (rehearse or (0 < ndev <= local_rank)) else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm
    E.init(dev_index)

    suffix = "" if args.dtype == "f32" else "-" + args.dtype
    name = f"ldpc/hip-{args.variant}{suffix}/{args.code}/{args.iters}"
    if args.rate not in ("", "none"):
        x, y = args.rate.split("/")
        name += f"/{x}/{y}"
    if args.path != "auto":
        os.environ["LDPC_HIP_PATH"] = args.path
    ecc = E.ECC(os.path.join(ROOT, "codes"), name, max_batch=args.batch)
    dec, sim, code = ecc.decoder, ecc.sim, ecc.code
    k, n_tx, N, Eg = ecc.message_length, ecc.codeword_length, code.N, code.E
    B = args.batch
    s_bytes = {"f32": 4, "f64": 8, "f16": 2}[args.dtype]

    # one explicit non-default stream for everything the library enqueues (a NULL handle would mean
    # "the context's own stream" for decode but the default stream for the frame source)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    sp = stream.cuda_stream
    nbuf = max(1, min(args.steps + args.warmup, 4))
    f16 = args.dtype == "f16"   # configs[3] "fp16 LLRs": the frame source writes fp16, the decoder reads fp16
    llr = [torch.empty((B, N), dtype=torch.float16 if f16 else torch.float32, device=dev) for _ in range(nbuf)]
    msg = [torch.empty((B, k), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    bits = torch.empty((B, N), dtype=torch.uint8, device=dev)
    iters_t = torch.empty((B,), dtype=torch.int32, device=dev)
    conv_t = torch.empty((B,), dtype=torch.uint8, device=dev)
    for i in range(nbuf):  # disjoint global frame ids per rank and buffer
        first = (rank * nbuf + i) * B
        sim.generate(args.seed, first, B, args.ebn0, llr[i].data_ptr(), msg[i].data_ptr(), sp, llr_f16=f16)
    torch.cuda.synchronize()

    def step(i):
        dec.decode_batch_dev(llr[i % nbuf].data_ptr(), bits.data_ptr(), B, args.iters, iters_t.data_ptr(), conv_t.data_ptr(), sp, llr_f16=f16)

    def barrier():
        if world > 1:
            dist.barrier()

    tally = torch.zeros(4, dtype=torch.int64, device=dev)  # frames, frame errors, bit errors, sum iters
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    dec.set_timing(True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    launches, kernel_ms = dec.kernel_time()
    dec.set_timing(False)
    # error statistics of the last step (outside the timed region)
    last = (args.warmup + args.steps - 1) % nbuf
    wrong = (bits[:, :k] != msg[last]).sum(dim=1)
    tally[0] = B
    tally[1] = (wrong > 0).sum()
    tally[2] = wrong.sum()
    tally[3] = iters_t.sum()
    from ecc_ldpc_amd import harness
    dist_mod = dist if world > 1 else None
    elapsed = harness.max_over_ranks(t1 - t0, dev, dist_mod)
    harness.all_reduce_tallies(tally, dist_mod)  # the path's only collective: 32 bytes over RCCL/xGMI
    frames_total = world * args.steps * B
    value = frames_total * k / elapsed / 1e6

    # ---- roofline of the dominant kernel (HIP events on the launch stream, live in this run)
    B_iter = (3 * Eg + 3 * N) * s_bytes
    B_cw = args.iters * B_iter + n_tx * s_bytes + (k + 7) // 8
    if dec.path == "fused":
        bytes_per_launch = B * B_cw
        bytes_note = "B_cw * frames per launch"
    else:
        bytes_per_launch = B * (2 * Eg + N) * s_bytes
        bytes_note = "(2E+N)*s * frames per launch (check-node kernel's share of B_iter)"
    avg_ms = kernel_ms / max(launches, 1)
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if launches else 0.0
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": measured_traffic(args, dec, B),
                "kernel": dec.kernel_name, "launches": launches, "avg_launch_ms": round(avg_ms, 4),
                "algorithmic_bytes_per_launch": bytes_per_launch, "bytes_model": bytes_note}
    if dec.path == "fused":
        # the fused kernels keep the BP state on-chip, so `frac` (the contract's HBM byte model) exceeds 1; what binds
        # them is VALU issue -- figures from the committed PMC passes of this workload, when there are any
        roofline["binding_resource"] = onchip_note(args)

    out = None
    if rank == 0:
        t = tally.tolist()
        out = {
            "metric": "decoded info Mbit/s @ 50 BP iters, jpl.4096.4.5, Eb/N0=2 dB",
            "value": round(value, 2), "unit": "Mbit/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.code} rate {args.rate} ({k},{n_tx}) {args.variant} flooding BP, {args.iters} iters, "
                                   f"Eb/N0={args.ebn0} dB, {B} frames/GPU/step", "code_name": ecc.name, "path": dec.path,
                       "batch_per_gpu": B, "parallelism": f"frames sharded over {world} GPU(s), tallies all-reduced"},
            **({"rehearsal": "ranks share GPUs over gloo; not a measurement"} if rehearse else {}),
            "roofline": roofline,
            "hbm_roofline_mbit_s": round(HBM_PEAK_GBS * 1e9 / B_cw * k / 1e6, 1),
            "frac_of_hbm_roofline_throughput": round(value / world / (HBM_PEAK_GBS * 1e9 / B_cw * k / 1e6), 4),
            "ber": t[2] / max(t[0] * k, 1), "fer": t[1] / max(t[0], 1), "mean_iters": t[3] / max(t[0], 1),
        }
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(args, ecc, llr[0], value)
        print(json.dumps(out), flush=True)
    # release the device objects in a known order before interpreter teardown
    torch.cuda.synchronize()
    ecc.close()
    del llr, msg, bits, iters_t, conv_t, tally
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def measured_traffic(args, dec, B):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/*_pmc.json: FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs of this same
    command), scaled by frames per launch; None when no profile of this configuration is committed."""
    if not (dec.path == "fused" and args.variant == "minsum" and args.dtype == "f32"):
        return None
    tag = {"jpl.4096.4.5": "r01_final_jpl4096_f32_minsum", "jpl.1024.4.5": "r01_final_jpl1024_f32_minsum"}.get(args.code)
    if tag is None:
        return None
    try:
        prof = json.load(open(os.path.join(ROOT, "profiles", tag + "_pmc.json")))
        h = prof["hbm_bytes_per_launch"]   # collected at 65536 frames per launch (tools/profile.sh)
        fetch = h.get("FETCH_SIZE_corrected_bytes", 2 * h["FETCH_SIZE_raw_bytes"])   # gfx950: raw FETCH_SIZE is 1/2 of the bytes
        return int((fetch + h["WRITE_SIZE_bytes"]) * B / 65536)
    except Exception:
        return None


def onchip_note(args):
    tag = {("jpl.4096.4.5", "minsum"): "r01_final_jpl4096_f32_minsum", ("jpl.1024.4.5", "minsum"): "r01_final_jpl1024_f32_minsum",
           ("jpl.4096.4.5", "tanh"): "r01_final_jpl4096_f32_tanh", ("1920.1280.3.303", "tanh"): "r01_final_mackay_f32_tanh"}.get((args.code, args.variant))
    note = {"bound": "valu issue + LDS round trips (state on-chip; HBM carries the LLRs in and the bits out only)"}
    try:
        m = json.load(open(os.path.join(ROOT, "profiles", tag + "_pmc.json")))["per_dispatch_mean"]
        gui = m["GRBM_GUI_ACTIVE"] / 8          # summed over the 8 XCDs
        note.update({"source": f"profiles/{tag}_pmc.json",
                     "valu_issue_interval_clk_per_simd": round(gui * 1024 / m["SQ_INSTS_VALU"], 2),
                     "waves_per_cu": round(m["SQ_WAVE_CYCLES"] * 4 / gui / 256, 1),
                     "lds_busy": round(m["SQ_LDS_IDX_ACTIVE"] / (gui * 256), 2),
                     "lds_bank_conflict_share": round(m["SQ_LDS_BANK_CONFLICT"] / max(m["SQ_LDS_IDX_ACTIVE"], 1), 2)})
    except Exception:
        pass
    return note


def host_cores():
    """CPU threads this process may really use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except Exception:
            continue
    env = os.environ.get("LDPC_CPU_THREADS")
    return int(env) if env else n


def cpu_baseline(args, ecc, llr_dev, gpu_value):
    """The CPU restatement of the reference decoder (oracle/, kind "port": the Haskell itself cannot be
    built here) timed on this box's host cores on a bounded sample of the SAME frames."""
    import numpy as np
    from oracle import oracle
    code = ecc.code
    rp, ci = code.csr()
    g = oracle.Graph(rp, ci, code.N)
    cores = host_cores()
    variant = "min" if args.variant == "minsum" else "tanh"
    probe = min(llr_dev.shape[0], 2 * cores)
    x = llr_dev[:probe].cpu().numpy().astype(np.float64)
    t0 = time.perf_counter()
    oracle.decode_batch(g, variant, args.iters, x, nthreads=cores)
    dt = time.perf_counter() - t0
    n = int(max(probe, min(llr_dev.shape[0], probe * args.cpu_seconds / max(dt, 1e-3))))
    n = max(cores, n // cores * cores)
    x = llr_dev[:n].cpu().numpy().astype(np.float64)
    t0 = time.perf_counter()
    oracle.decode_batch(g, variant, args.iters, x, nthreads=cores)
    dt = time.perf_counter() - t0
    v = n * ecc.message_length / dt / 1e6
    return {"value": round(v, 4), "unit": "Mbit/s", "cores": cores, "kind": "port",
            "sample": f"first {n} frames of the step-0 batch, {args.iters} iters, double precision, {cores} threads, {dt:.1f} s",
            "gpu_over_cpu": round(gpu_value / v, 1) if v > 0 else None}


if __name__ == "__main__":
    main()  # normal interpreter exit: rocprofv3 writes its output from exit handlers
