#!/usr/bin/env python3
"""bench.py -- decoded information Mbit/s of the LDPC BP decode path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  N > 1, either way: `python bench.py --gpus N ...` starts its own N child ranks (ecc_ldpc_amd/launch.py; the parent never
  touches the GPU), or `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...`

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
N_SIMD, CLOCK_HZ = 256 * 4, 2.4e9          # 256 CUs x 4 SIMDs, peak engine clock
VALU_PEAK = N_SIMD * CLOCK_HZ / 2          # wave-instructions / s: a full-rate VALU instruction issues every 2 clk per SIMD


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--code", default="jpl.4096.4.5")
    ap.add_argument("--rate", default="4/5")
    ap.add_argument("--variant", default="minsum", choices=["minsum", "tanh"])
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64", "f16", "f16pk"],
                    help="f16: fp16 LLRs, f32 arithmetic; f16pk: fp16 LLRs AND fp16 arithmetic, two frames per lane (BASELINE configs[3])")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--ebn0", type=float, default=2.0)
    ap.add_argument("--batch", type=int, default=65536, help="frames per GPU per step")
    ap.add_argument("--path", default="auto", choices=["auto", "flood", "fused"])
    ap.add_argument("--schedule", default="flooding", choices=["flooding", "layered"], help="layered: extension (BASELINE configs[4])")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the cpu_baseline sample (0 = skip)")
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5EEDC0DE)
    ap.add_argument("--fp16-leg", type=int, default=1, help="0: skip the extra, separately labelled measurements of the same workload (packed-fp16 decoder; layered schedule f32 / packed fp16) "
                    "(BASELINE configs[3]; reported next to the f32 headline as `fp16_packed`, never as `value`)")
    ap.add_argument("--live-traffic", type=int, default=-1, help="1: measure roofline.traffic in this run (two short child runs of this workload under "
                    "rocprofv3 --pmc, N = 1 only); 0: take it from the committed PMC pass; default: 1 for a full run, 0 when --cpu-seconds 0 trims it")
    ap.add_argument("--proof", type=int, default=1, help="0: skip the untimed proof-of-work sample (tools/profile.sh does, so that the "
                    "kernel-trace average covers full-size launches only)")
    args = ap.parse_args()
    if args.live_traffic < 0:
        args.live_traffic = 1 if args.cpu_seconds > 0 else 0

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python3 bench.py --gpus N` by itself: this parent starts one child process per GPU and relays rank 0's line.
        # It must happen HERE, before torch or the HIP library is imported -- the parent never initialises the GPU
        # and nothing is re-exec'ed (ecc_ldpc_amd/launch.py).
        raise SystemExit(self_launch(args.gpus))

    import numpy as np
    import torch
    import ecc_ldpc_amd as E

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # LDPC_BENCH_REHEARSE=1: run the N > 1 code path on a box with fewer GPUs than ranks (ranks share devices,
    # gloo instead of RCCL).  For checking the launch/sharding/tally logic only; its number is not a result.
    rehearse = os.environ.get("LDPC_BENCH_REHEARSE") == "1"
    ndev = torch.cuda.device_count()
    # (a launcher that gives every rank its own HIP_VISIBLE_DEVICES shows each rank ONE device: index 0)
    dev_index = local_rank % ndev if (rehearse or (0 < ndev <= local_rank)) else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    # LDPC_BENCH_FORCE_DIST=1: initialise the process group and run the collectives even with ONE rank -- the RCCL calls of
    # the N > 1 path on a 1-GPU box (tests/test_bench_contract_gpu.py)
    force_dist = os.environ.get("LDPC_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm
    E.init(dev_index)

    suffix = "" if args.dtype == "f32" else "-" + args.dtype
    name = f"ldpc/hip-{args.variant}{'-layered' if args.schedule == 'layered' else ''}{suffix}/{args.code}/{args.iters}"
    if args.rate not in ("", "none"):
        x, y = args.rate.split("/")
        name += f"/{x}/{y}"
    if args.path != "auto":
        os.environ["LDPC_HIP_PATH"] = args.path
    ecc = E.ECC(os.path.join(ROOT, "codes"), name, max_batch=args.batch)
    dec, sim, code = ecc.decoder, ecc.sim, ecc.code
    k, n_tx, N, Eg = ecc.message_length, ecc.codeword_length, code.N, code.E
    B = args.batch
    s_bytes = {"f32": 4, "f64": 8, "f16": 2, "f16pk": 2}[args.dtype]
    B_cw = args.iters * (3 * Eg + 3 * N) * s_bytes + n_tx * s_bytes + (k + 7) // 8   # SURVEY.md section 8d, at max_iters

    # one explicit non-default stream for everything the library enqueues (a NULL handle would mean
    # "the context's own stream" for decode but the default stream for the frame source)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    sp = stream.cuda_stream
    nbuf = max(1, min(args.steps + args.warmup, 4))
    f16 = args.dtype in ("f16", "f16pk")   # configs[3] "fp16 LLRs": the frame source writes fp16, the decoder reads fp16
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
        if dist is not None:
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
    dist_mod = dist
    elapsed = harness.max_over_ranks(t1 - t0, dev, dist_mod)
    rank_seconds = harness.per_rank(t1 - t0, dev, dist_mod)     # one entry per rank that took part in the collective
    ones = torch.ones(1, dtype=torch.int64, device=dev)
    harness.all_reduce_tallies(ones, dist_mod)                 # sum of ones over the data-path backend (RCCL unless rehearsing)
    harness.all_reduce_tallies(tally, dist_mod)  # the path's only collective: 32 bytes over RCCL/xGMI
    frames_total = world * args.steps * B
    value = frames_total * k / elapsed / 1e6

    # ---- iterations the timed steps really ran (early exit): one untimed pass per distinct input buffer
    sum_iters_buf = []
    for i in range(nbuf):
        step(i)
        torch.cuda.synchronize()
        sum_iters_buf.append(int(iters_t.sum().item()))
    turns_timed = sum(sum_iters_buf[(args.warmup + i) % nbuf] for i in range(args.steps))   # frame-turns of this rank
    roofline, roofline_hbm = rooflines(args, dec, B, Eg, N, n_tx, k, s_bytes, launches, kernel_ms, turns_timed, live=(rank == 0))
    pow_obj = proof_of_work(args, E, ecc, dec, sim, llr[0], msg[0], bits, iters_t, conv_t, sp, f16) if (rank == 0 and args.proof) else None

    out = None
    if rank == 0:
        t = tally.tolist()
        out = {
            "metric": f"decoded info Mbit/s @ {args.iters} BP iters, {args.code}, Eb/N0={args.ebn0:g} dB",
            "value": round(value, 2), "unit": "Mbit/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16" if args.dtype == "f16pk" else args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.code} rate {args.rate} ({k},{n_tx}) {args.variant} {args.schedule} BP, {args.iters} iters, "
                                   f"Eb/N0={args.ebn0} dB, {B} frames/GPU/step", "code_name": ecc.name, "path": dec.path,
                       "batch_per_gpu": B, "parallelism": f"frames sharded over {world} GPU(s), tallies all-reduced"},
            **({"rehearsal": "ranks share GPUs over gloo; not a measurement"} if rehearse else {}),
            "ranks_seen": int(ones.item()), "per_rank_ms_per_step": [round(s / args.steps * 1e3, 3) for s in rank_seconds],
            "launcher": os.environ.get("LDPC_BENCH_LAUNCHER", "external" if "WORLD_SIZE" in os.environ else "none"),
            "collective_backend": (dist.get_backend() if dist is not None else None),
            "roofline": roofline,
            **({"roofline_hbm_model": roofline_hbm} if roofline_hbm else {}),
            "proof_of_work": pow_obj,
            "hbm_roofline_mbit_s": round(HBM_PEAK_GBS * 1e9 / B_cw * k / 1e6, 1),
            "frac_of_hbm_roofline_throughput": round(value / world / (HBM_PEAK_GBS * 1e9 / B_cw * k / 1e6), 4),
            "ber": t[2] / max(t[0] * k, 1), "fer": t[1] / max(t[0], 1), "mean_iters": t[3] / max(t[0], 1),
        }
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(args, ecc, llr[0], value)
        if world == 1 and args.dtype == "f32" and args.variant == "minsum" and args.schedule == "flooding" and args.path == "auto" and args.fp16_leg:
            out["fp16_packed"] = fp16_packed_leg(args, E, torch, dev, sp, B)
            # the row-layered schedule (an extension: the reference has flooding only) on the same code, on-chip, f32 and packed fp16,
            # at the headline Eb/N0 and in the waterfall -- labelled measurements next to `value`, never instead of it
            out["layered_extension"] = [layered_leg(args, E, torch, dev, sp, B, f16pk, db) for f16pk in (False, True) for db in sorted({args.ebn0, 3.0})]
            # BASELINE configs[4]: the DVB-S2-shaped n = 64 800 code (a SYNTHETIC matrix of that shape: the ETSI tables are not available),
            # layered min-sum with early termination -- lam on-chip in fp16, row records streamed from HBM (csrc/layered_lds.hip, r04)
            out["long_code_layered"] = long_code_leg(args, E, torch, dev, sp)
            # BASELINE configs[1] (jpl.1024.4.5 min-sum f32) and configs[2] (1920.1280.3.303 tanh f32, the 1-4 dB sweep): labelled measurements too
            if args.code == "jpl.4096.4.5":
                # (configs[0] is the reference's own CPU-runnable plumbing case, the toy moon.7.13 code with the tanh rule and 20 turns: here it
                #  goes through the generic on-chip kernel like any other H)
                out["configs0_moon_tanh"] = other_config_leg(args, E, torch, dev, sp, B, "moon.7.13", "none", "tanh", [args.ebn0, 6.0], iters=20)
                out["configs1_jpl1024_minsum"] = other_config_leg(args, E, torch, dev, sp, B, "jpl.1024.4.5", "4/5", "minsum", [args.ebn0])
                out["configs2_mackay_tanh_sweep"] = other_config_leg(args, E, torch, dev, sp, B, "1920.1280.3.303", "none", "tanh", [1.0, 2.0, 3.0, 4.0], live_db=1.0)
        print(json.dumps(out), flush=True)
    # release the device objects in a known order before interpreter teardown
    torch.cuda.synchronize()
    ecc.close()
    del llr, msg, bits, iters_t, conv_t, tally
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def layered_leg(args, E, torch, dev, sp, B, f16pk, ebn0):
    name = f"ldpc/hip-minsum-layered{'-f16pk' if f16pk else ''}/{args.code}/{args.iters}"
    if args.rate not in ("", "none"):
        x, y = args.rate.split("/")
        name += f"/{x}/{y}"
    try:
        ecc = E.ECC(os.path.join(ROOT, "codes"), name, max_batch=B)
    except E.LdpcError as e:
        return {"code_name": name, "error": str(e)}
    dec, sim, k, N = ecc.decoder, ecc.sim, ecc.message_length, ecc.code.N
    llr = torch.empty((B, N), dtype=torch.float16 if f16pk else torch.float32, device=dev)
    msg = torch.empty((B, k), dtype=torch.uint8, device=dev)
    bits = torch.empty((B, N), dtype=torch.uint8, device=dev)
    its = torch.empty((B,), dtype=torch.int32, device=dev)
    sim.generate(args.seed, 0, B, ebn0, llr.data_ptr(), msg.data_ptr(), sp, llr_f16=f16pk)
    step = lambda: dec.decode_batch_dev(llr.data_ptr(), bits.data_ptr(), B, args.iters, its.data_ptr(), None, sp, llr_f16=f16pk)
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    wrong = (bits[:, :k] != msg).sum(dim=1)
    res = {"code_name": name, "ebn0_db": ebn0, "value": round(args.steps * B * k / dt / 1e6, 2), "unit": "Mbit/s", "ms_per_step": round(dt / args.steps * 1e3, 3),
           "path": dec.path, "kernel": dec.kernel_name, "ber": float(wrong.sum().item()) / (B * k), "fer": float((wrong > 0).sum().item()) / B,
           "mean_sweeps": float(its.float().mean().item()),
           "checked_by": "tests/test_pk16_gpu.py (emulation, bit for bit)" if f16pk else "tests/test_layered_fused_gpu.py (HBM layered kernel bit for bit; oracle_decode_layered hard bits)"}
    ecc.close()
    del llr, msg, bits, its
    return res


def other_config_leg(args, E, torch, dev, sp, B, code, rate, variant, dbs, live_db=None, iters=None):
    """BASELINE.json configs[1] / configs[2] next to the headline: another shipped matrix through the same entry points, f32, flooding, the
    same number of timed steps per Eb/N0 point; HIP-event kernel time; BER / FER against the transmitted messages (the all-zero word
    where the matrix ships without a generator).  `live_db`: the point whose HBM traffic and pipe occupancy are measured in the run."""
    if iters is not None:
        import types
        args = types.SimpleNamespace(**vars(args))
        args.iters = iters
    name = f"ldpc/hip-{'minsum' if variant == 'minsum' else 'tanh'}/{code}/{args.iters}"
    if rate not in ("", "none"):
        x, y = rate.split("/")
        name += f"/{x}/{y}"
    try:
        ecc = E.ECC(os.path.join(ROOT, "codes"), name, max_batch=B)
    except E.LdpcError as e:
        return {"code_name": name, "error": str(e)}
    dec, sim, k, N = ecc.decoder, ecc.sim, ecc.message_length, ecc.code.N
    llr = torch.empty((B, N), dtype=torch.float32, device=dev)
    llr_b = torch.empty((B, N), dtype=torch.float32, device=dev)     # a second batch of other frames: consecutive launches never read the same bytes
    msg = torch.zeros((B, k), dtype=torch.uint8, device=dev)
    msg_b = torch.zeros((B, k), dtype=torch.uint8, device=dev)
    bits = torch.empty((B, N), dtype=torch.uint8, device=dev)
    its = torch.empty((B,), dtype=torch.int32, device=dev)
    has_g = ecc.sim.encoder != "none"
    steps = max(1, min(args.steps, 4))
    points = []
    for db in dbs:
        sim.generate(args.seed, B, B, db, llr_b.data_ptr(), msg_b.data_ptr() if has_g else None, sp)
        sim.generate(args.seed, 0, B, db, llr.data_ptr(), msg.data_ptr() if has_g else None, sp)
        step = lambda i=0: dec.decode_batch_dev((llr_b if i & 1 else llr).data_ptr(), bits.data_ptr(), B, args.iters, its.data_ptr(), None, sp)
        step(1)
        torch.cuda.synchronize()
        dec.set_timing(True)
        t0 = time.perf_counter()
        for i in range(steps):
            step(steps - 1 - i)                      # (alternating, ending on the first batch: the tallies below are its)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        launches, kernel_ms = dec.kernel_time()
        dec.set_timing(False)
        wrong = (bits[:, :k] != msg).sum(dim=1)
        pt = {"ebn0_db": db, "value": round(steps * B * k / dt / 1e6, 2), "unit": "Mbit/s", "ms_per_step": round(dt / steps * 1e3, 3),
              "avg_launch_ms": round(kernel_ms / max(launches, 1), 4), "ber": float(wrong.sum().item()) / (B * k), "fer": float((wrong > 0).sum().item()) / B,
              "mean_iters": float(its.float().mean().item())}
        ent = isa_entry(dec.kernel_name, dec.code, "min" if variant == "minsum" else "tanh")
        threads, fpw = dec.kernel_geometry
        if ent is not None and kernel_ms and threads and fpw:
            loops = ent["loops"]
            valu = sum(lp["hot_turn"]["units"].get("valu", 0) for lp in loops) / len(loops)
            pt["valu_frac"] = round(valu * (threads / 64.0 / fpw) * float(its.sum().item()) * steps / (kernel_ms * 1e-3) / VALU_PEAK, 4)
        points.append(pt)
    res = {"code_name": name, "frames": B, "steps": steps, "path": dec.path, "kernel": dec.kernel_name, "points": points,
           "checked_by": "tests/test_fused_gpu.py, tests/test_fused_csr_gpu.py, tests/test_golden.py: hard bits, flags and iteration counts against the oracle"}
    kname = dec.kernel_name
    ecc.close()
    del llr, llr_b, msg, msg_b, bits, its
    torch.cuda.empty_cache()
    if live_db is not None:
        import types
        a2 = types.SimpleNamespace(**vars(args))
        a2.code, a2.rate, a2.variant, a2.dtype, a2.path, a2.schedule, a2.ebn0 = code, rate, variant, "f32", "auto", "flooding", live_db
        holder = types.SimpleNamespace(kernel_name=kname)
        traffic, tsrc = live_traffic(a2, holder, B)
        if traffic is not None:
            res.update({"traffic_at_db": live_db, "traffic": traffic, "traffic_source": tsrc})
            busy = live_pipe_busy(a2, holder, B)
            if busy:
                res["counters"] = busy
    return res


def long_code_leg(args, E, torch, dev, sp, B=16384, code="dvbs2like.64800.1.2"):
    name = f"ldpc/hip-minsum-layered-f16/{code}/{args.iters}"
    try:
        ecc = E.ECC(os.path.join(ROOT, "codes"), name, max_batch=B)
    except E.LdpcError as e:
        return {"code_name": name, "error": str(e)}
    dec, sim, k, N, M = ecc.decoder, ecc.sim, ecc.message_length, ecc.code.N, ecc.code.M
    llr = torch.empty((B, N), dtype=torch.float16, device=dev)
    bits = torch.empty((B, N), dtype=torch.uint8, device=dev)
    its = torch.empty((B,), dtype=torch.int32, device=dev)
    conv = torch.empty((B,), dtype=torch.uint8, device=dev)
    sim.generate(args.seed, 0, B, args.ebn0, llr.data_ptr(), None, sp, llr_f16=True)     # (no generator: frames are the all-zero codeword + noise)
    step = lambda: dec.decode_batch_dev(llr.data_ptr(), bits.data_ptr(), B, args.iters, its.data_ptr(), conv.data_ptr(), sp, llr_f16=True)
    step()
    torch.cuda.synchronize()
    dec.set_timing(True)
    steps = max(1, min(args.steps, 4))
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    launches, kernel_ms = dec.kernel_time()
    dec.set_timing(False)
    wrong = bits[:, :k].ne(0).sum(dim=1)            # (the transmitted codeword is all-zero)
    sweeps = float(its.double().sum().item())
    alg = steps * (sweeps * 24 * M - B * 12 * M + B * (N * 2 + N))     # records read + written per sweep (the first sweep writes only), LLRs in, bits out
    res = {"metric": f"decoded info Mbit/s, {name}, Eb/N0={args.ebn0:g} dB", "value": round(steps * B * k / dt / 1e6, 2), "unit": "Mbit/s", "frames": B, "steps": steps,
           "ms_per_step": round(dt / steps * 1e3, 3), "avg_launch_ms": round(kernel_ms / max(launches, 1), 4), "kernel": dec.kernel_name, "path": dec.path,
           "threads_per_workgroup": dec.kernel_geometry[0], "ber": float(wrong.sum().item()) / (B * k), "fer": float((wrong > 0).sum().item()) / B,
           "converged_frac": float(conv.float().mean().item()), "mean_sweeps": sweeps / B,
           "roofline": {"bound": "hbm", "achieved": round(alg / (kernel_ms * 1e-3) / 1e9, 1) if kernel_ms else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(alg / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if kernel_ms else None,
                        "bytes_model": "sum_frames(sweeps_f)*24M - frames*12M + frames*(2N + N): one 12-byte record per row read + written per sweep, fp16 LLRs in, bits out"},
           "matrix": "SYNTHETIC, DVB-S2 rate-1/2 normal-frame shape (tools/gen_dvbs2_like.py); frames = all-zero codeword + AWGN",
           "checked_by": "tests/test_layered_gpu.py: bit-exact with oracle/emulate_f16.py decode_minsum_f16_layered"}
    name_of_kernel = dec.kernel_name
    ecc.close()
    del llr, bits, its, conv
    torch.cuda.empty_cache()
    # HBM bytes per launch and vector-pipe occupancy of THIS workload, measured now (child runs under rocprofv3 --pmc, as for the headline)
    import types
    a2 = types.SimpleNamespace(**vars(args))
    a2.code, a2.rate, a2.variant, a2.dtype, a2.path, a2.schedule = code, "none", "minsum", "f16", "auto", "layered"
    holder = types.SimpleNamespace(kernel_name=name_of_kernel)
    traffic, tsrc = live_traffic(a2, holder, B)
    if traffic is not None:
        res["roofline"].update({"traffic": traffic, "traffic_source": tsrc, "traffic_over_algorithmic": round(traffic / (alg / steps), 3)})
        busy = live_pipe_busy(a2, holder, B)
        if busy:
            res["roofline"]["counters"] = busy
    return res


def fp16_packed_leg(args, E, torch, dev, sp, B):
    """BASELINE configs[3] on the same workload: fp16 LLRs, ARITHMETIC in packed fp16, two frames per lane (LDPC_F16PK,
    csrc/fused_pk16_body.h) -- the same frame ids, the same number of timed steps, HIP-event kernel time and wall time.  A second,
    separately labelled measurement: `value` above stays the f32 decoder's."""
    name = f"ldpc/hip-minsum-f16pk/{args.code}/{args.iters}"
    if args.rate not in ("", "none"):
        x, y = args.rate.split("/")
        name += f"/{x}/{y}"
    try:
        ecc = E.ECC(os.path.join(ROOT, "codes"), name, max_batch=B)
    except E.LdpcError as e:
        return {"error": str(e)}
    dec, sim, k, N = ecc.decoder, ecc.sim, ecc.message_length, ecc.code.N
    llr = torch.empty((B, N), dtype=torch.float16, device=dev)
    msg = torch.empty((B, k), dtype=torch.uint8, device=dev)
    bits = torch.empty((B, N), dtype=torch.uint8, device=dev)
    its = torch.empty((B,), dtype=torch.int32, device=dev)
    sim.generate(args.seed, 0, B, args.ebn0, llr.data_ptr(), msg.data_ptr(), sp, llr_f16=True)
    step = lambda: dec.decode_batch_dev(llr.data_ptr(), bits.data_ptr(), B, args.iters, its.data_ptr(), None, sp, llr_f16=True)
    step()
    torch.cuda.synchronize()
    dec.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    launches, kernel_ms = dec.kernel_time()
    dec.set_timing(False)
    wrong = (bits[:, :k] != msg).sum(dim=1)
    threads, fpw = dec.kernel_geometry
    res = {"metric": f"decoded info Mbit/s, {name}", "value": round(args.steps * B * k / dt / 1e6, 2), "unit": "Mbit/s", "dtype": "f16 (arithmetic and LLRs), two frames per lane",
           "steps": args.steps, "ms_per_step": round(dt / args.steps * 1e3, 3), "avg_launch_ms": round(kernel_ms / max(launches, 1), 4), "kernel": dec.kernel_name,
           "threads_per_workgroup": threads, "frames_per_workgroup": fpw, "ber": float(wrong.sum().item()) / (B * k), "fer": float((wrong > 0).sum().item()) / B,
           "mean_iters": float(its.float().mean().item()),
           "checked_by": "tests/test_pk16_gpu.py: bit-exact with oracle/emulate_f16.py decode_minsum_pk16 (the reference has no fp16 decoder)"}
    ent = isa_entry(dec.kernel_name)
    if ent is not None and kernel_ms:
        loops = ent["loops"]
        valu = sum(lp["hot_turn"]["units"].get("valu", 0) for lp in loops) / len(loops)
        clk = sum(lp["hot_turn"]["valu_cost_weighted_clk"] for lp in loops) / len(loops)
        turns = float(its.sum().item()) * args.steps
        waves_per_frame = threads / 64.0 / fpw
        res.update({"valu_instr_per_wave_turn": round(valu, 1), "waves_per_frame": waves_per_frame,
                    "valu_frac": round(valu * waves_per_frame * turns / (kernel_ms * 1e-3) / VALU_PEAK, 4),
                    "valu_pipe_busy_frac": round(clk * waves_per_frame * turns / (N_SIMD * CLOCK_HZ * kernel_ms * 1e-3), 4)})
    kname = dec.kernel_name
    ecc.close()
    del llr, msg, bits, its
    torch.cuda.empty_cache()
    # HBM bytes per launch and vector-pipe occupancy of THIS workload, measured now (child runs under rocprofv3 --pmc, as for the headline)
    import types
    a2 = types.SimpleNamespace(**vars(args))
    a2.dtype = "f16pk"
    holder = types.SimpleNamespace(kernel_name=kname)
    traffic, tsrc = live_traffic(a2, holder, B)
    if traffic is not None:
        res.update({"traffic": traffic, "traffic_source": tsrc})
        busy = live_pipe_busy(a2, holder, B)
        if busy:
            res["counters"] = busy
    return res


def self_launch(n):
    """Parent of an N-rank run (see main): builds the library once if this is a fresh checkout (so the ranks do not
    race for it; _lib.py's file lock stays as the second line), then one child per rank with the same arguments."""
    import subprocess
    from ecc_ldpc_amd import launch     # standard library only: no torch, no libldpc_hip.so
    so = os.path.join(ROOT, "ecc_ldpc_amd", "libldpc_hip.so")
    if not os.path.exists(so) and "LDPC_SO" not in os.environ:
        rc = subprocess.call([sys.executable, os.path.join(ROOT, "ecc_ldpc_amd", "build.py")], stdout=sys.stderr)
        if rc != 0:
            return rc
    return launch.launch_ranks(n, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                               timeout=float(os.environ.get("LDPC_BENCH_LAUNCH_TIMEOUT", "1500")))


def proof_of_work_layered(args, E, ecc, dec, sim, llr_t, msg_t, bits, iters_t, conv_t, sp, f16, n):
    """The layered schedule has one implementation, so the check is by invariant, with a DIFFERENT kernel as the judge:
    every frame reported converged must be a codeword -- its hard bits, fed back as LLRs to a flooding-path context with
    0 turns allowed, must come back 'syndrome zero' from flood_cn_kernel -- a frame reported failed must carry the
    channel's hard decisions (the reference's rule, Orig.hs:70), and decoding must remove bit errors."""
    import torch
    k = ecc.message_length
    out = {"sample_frames": n, "checked_against": "syndrome kernel of the flooding path on the decoder's output; channel decisions for failed frames", "points": []}
    chk = E.Decoder(ecc.code, "min", "f32", n, path="flood")
    cb = torch.empty_like(bits[:n])
    ci = torch.empty_like(iters_t[:n])
    cc = torch.empty_like(conv_t[:n])
    for db in (args.ebn0, args.ebn0 + 1.6):
        sim.generate(args.seed, 1 << 40, n, db, llr_t.data_ptr(), msg_t.data_ptr(), sp, llr_f16=f16)
        dec.decode_batch_dev(llr_t.data_ptr(), bits.data_ptr(), n, args.iters, iters_t.data_ptr(), conv_t.data_ptr(), sp, llr_f16=f16)
        torch.cuda.synchronize()
        as_llr = (bits[:n].to(torch.float32) * 2.0 - 1.0).contiguous()
        chk.decode_batch_dev(as_llr.data_ptr(), cb.data_ptr(), n, 0, ci.data_ptr(), cc.data_ptr(), sp)
        torch.cuda.synchronize()
        conv = conv_t[:n].bool()
        hard_in = (llr_t[:n].float() > 0).to(torch.uint8)
        raw = (hard_in[:, :k] != msg_t[:n]).sum().item()
        dec_err = (bits[:n, :k] != msg_t[:n]).sum().item()
        it = iters_t[:n].cpu().numpy()
        out["points"].append({
            "ebn0_db": db, "converged_frames_are_codewords": bool(cc.bool()[conv].all().item()),
            "failed_frames_return_channel_decisions": bool((bits[:n][~conv] == hard_in[~conv]).all().item()),
            "converged_frac": round(float(conv.float().mean().item()), 4), "mean_sweeps": round(float(it.mean()), 2),
            "distinct_sweep_counts": int(len(set(it.tolist()))), "channel_bit_errors": int(raw), "decoded_bit_errors": int(dec_err)})
    chk.close()
    out["ok"] = all(p["converged_frames_are_codewords"] and p["failed_frames_return_channel_decisions"] for p in out["points"])
    return out


PROFILE_TAGS = {   # committed rocprofv3 PMC passes (tools/profile.sh): (code, variant, dtype, kernel family) -> (tag, frames per launch profiled)
    ("jpl.4096.4.5", "minsum", "f32", "fused"): ("jpl4096_f32_minsum", 65536), ("jpl.1024.4.5", "minsum", "f32", "fused"): ("jpl1024_f32_minsum", 65536),
    ("jpl.4096.4.5", "tanh", "f32", "fused"): ("jpl4096_f32_tanh", 16384), ("1920.1280.3.303", "tanh", "f32", "fused"): ("mackay_f32_tanh", 65536),
    ("jpl.4096.4.5", "minsum", "f16pk", "fused"): ("jpl4096_f16pk_minsum", 65536),
    ("jpl.4096.4.5", "minsum", "f32", "flood_qc"): ("floodqc_jpl4096_f32_minsum", 16384),
    ("dvbs2like.64800.1.2", "minsum", "f32", "layered_qc"): ("dvbs2like_layered_f32_minsum", 32768),
    ("dvbs2like.64800.1.2", "minsum", "f16", "layered_lds"): ("dvbs2like_layered_f16_minsum", 32768),   # r04: lam on-chip, records streamed
    # the on-chip layered kernels (family "fused_layered": another kernel than the flooding one of the same code and type)
    ("jpl.4096.4.5", "minsum", "f32", "fused_layered"): ("jpl4096_layered_f32_minsum", 65536),
    ("jpl.4096.4.5", "minsum", "f16pk", "fused_layered"): ("jpl4096_layered_f16pk_minsum", 65536),
    ("1920.1280.A", "minsum", "f32", "fused"): ("1920A_f32_minsum", 16384)}


_LIVE = {}      # kernel name -> per-launch means of the counters measured in this run (live_counters)


def live_counters(args, dec, B):
    """Per-launch means of hardware counters of the dominant kernel MEASURED IN THIS RUN: the same workload three more times, a few steps
    each, as child processes under `rocprofv3 --pmc ...` -- FETCH_SIZE and WRITE_SIZE in passes of their own, as the guide's HBM section
    prescribes, then the shader counters that say how busy the vector pipes were; nothing else is traced in those runs.  Counters cannot
    be read from inside the process that is being timed; the children run after the timed region, one at a time, while this process only
    waits.  -> {counter: mean, "launches": n} or None: no rocprofv3, a child failed or took too long, more than one rank, this IS such a
    child, or this process is itself being profiled."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    from collections import defaultdict
    if not args.live_traffic or os.environ.get("LDPC_BENCH_CHILD") or args.gpus != 1 or int(os.environ.get("WORLD_SIZE", "1")) > 1:
        return None
    key = dec.kernel_name.strip().rstrip(",").strip()
    if key in _LIVE:
        return _LIVE[key]
    exe = shutil.which("rocprofv3")
    if not exe:
        return None
    if "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None                # this process is being profiled itself: no profiler inside a profiler
    child = [sys.executable or "python3", os.path.abspath(__file__), "--gpus", "1", "--steps", "2", "--warmup", "1", "--code", args.code, "--rate", args.rate,
             "--variant", args.variant, "--dtype", args.dtype, "--iters", str(args.iters), "--ebn0", str(args.ebn0), "--batch", str(B), "--path", args.path,
             "--schedule", args.schedule, "--cpu-seconds", "0", "--seed", hex(args.seed), "--fp16-leg", "0", "--proof", "0", "--live-traffic", "0"]
    env = dict(os.environ, LDPC_BENCH_CHILD="1", TMPDIR="/tmp")
    for k in list(env):              # a child is a plain one-process run, whatever launched this one
        if k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "ROLE_RANK",
                 "ROLE_WORLD_SIZE", "ROLE_NAME") or k.startswith(("TORCHELASTIC_", "TORCH_NCCL_", "NCCL_ASYNC")):
            del env[k]
    got = {}
    for ctrs in (["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY"]):
        d = tempfile.mkdtemp(prefix="ldpc_pmc_", dir="/tmp")
        try:
            # (its own process group: on a time-out the profiler AND the run it started are ended, by that group's id -- nothing is left behind)
            pr = subprocess.Popen([exe, "--pmc"] + ctrs + ["--output-format", "csv", "-d", d, "-o", "pmc", "--"] + child, cwd="/tmp", env=env,
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = pr.wait(timeout=150)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(pr.pid, signal.SIGKILL)
                except OSError:
                    pass
                pr.wait()
                raise
            if rc != 0:
                raise RuntimeError(f"rocprofv3 exited with {rc}")
            per = defaultdict(float)     # (dispatch, counter) -> value summed over its dimensions (XCDs, SEs)
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if key in r["Kernel_Name"] and r["Counter_Name"] in ctrs:
                        per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
            for c in ctrs:
                v = [x for (_, cn), x in per.items() if cn == c]
                if not v:
                    raise RuntimeError(c)
                got[c] = sum(v) / len(v)
                got["launches"] = len(v)
        except Exception:
            if "FETCH_SIZE" in got and "WRITE_SIZE" in got:
                break                    # the byte counters are in: keep them, do without the shader counters
            _LIVE[key] = None
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    _LIVE[key] = got
    return got


def live_traffic(args, dec, B):
    """-> (HBM bytes per launch measured in this run, what it is) or (None, None); FETCH_SIZE doubled: gfx950 tallies 128-byte read
    requests at 64 (MI355X_MICROARCH.md, HBM / rocprofv3 section); the counters are in KB"""
    got = live_counters(args, dec, B)
    if not got:
        return None, None
    rd, wr = 2.0 * got["FETCH_SIZE"] * 1024.0, got["WRITE_SIZE"] * 1024.0
    return int(rd + wr), (f"measured in this run: child runs under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), mean over "
                          f"{got['launches']} launches of {B} frames; read side doubled (gfx950 counts 128-byte requests at 64): "
                          f"{int(rd)} B read + {int(wr)} B written")


def live_pipe_busy(args, dec, B):
    """-> {"valu_pipe_busy": ..., "waves_waiting": ...} from the shader counters of live_counters, or None.  SQ_ACTIVE_INST_VALU counts
    quad-cycles a wave had a VALU instruction in flight (every instruction rounded up to one), summed over the device: x 4 clk / 1024
    SIMDs against the launch's cycles (GRBM_GUI_ACTIVE is summed over the 8 XCDs).  Near or above 1 = saturated."""
    got = live_counters(args, dec, B)
    if not got or "SQ_ACTIVE_INST_VALU" not in got or not got.get("GRBM_GUI_ACTIVE"):
        return None
    cyc = got["GRBM_GUI_ACTIVE"] / 8.0
    return {"valu_pipe_busy": round(got["SQ_ACTIVE_INST_VALU"] * 4.0 / N_SIMD / cyc, 3),
            "waves_waiting": round(got["SQ_WAIT_ANY"] / got["SQ_WAVE_CYCLES"], 3) if got.get("SQ_WAVE_CYCLES") else None,
            "source": "measured in this run: child run under rocprofv3 --pmc SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY; "
                      "valu_pipe_busy = SQ_ACTIVE_INST_VALU x 4 clk / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs); the counter rounds every instruction "
                      "up to a quad-cycle, so ~1 means saturated"}


def committed_traffic(args, dec, B):
    """HBM bytes per launch of the dominant kernel from a COMMITTED rocprofv3 PMC pass (profiles/*_pmc.json:
    FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs of this same command, FETCH_SIZE doubled as the
    gfx950 guide prescribes), scaled by frames per launch -> (bytes, source file) or (None, None).  Counters cannot be
    read from inside the process; this value is therefore NOT measured in this run and says which file it is from.
    (For a kernel with early exit the scaling by frames assumes the profiled Eb/N0.)"""
    kname = dec.kernel_name
    family = "fused" if dec.path == "fused" else ("flood_qc" if "flood_qc_kernel" in kname else ("layered_qc" if "layered_qc_kernel" in kname else
                                                  ("layered_lds" if "layered_lds_kernel" in kname else None)))
    if family == "fused" and args.schedule == "layered":
        family = "fused_layered"
    ent = PROFILE_TAGS.get((args.code, args.variant, args.dtype, family))
    if ent is None:
        return None, None
    tag, frames = ent
    for rnd in ("r04_final_", "r04_", "r03_final_", "r02_final_", "r01_final_"):
        path = os.path.join(ROOT, "profiles", rnd + tag + "_pmc.json")
        try:
            prof = json.load(open(path))
            if family in ("flood_qc", "layered_qc") and family.replace("_qc", "_qc_kernel") not in prof["kernel"]:
                continue
            if family == "layered_lds" and "layered_lds_kernel" not in prof["kernel"]:
                continue
            if family == "fused_layered" and "layered" not in prof["kernel"]:
                continue
            h = prof["hbm_bytes_per_launch"]
            fetch = h.get("FETCH_SIZE_corrected_bytes", 2 * h["FETCH_SIZE_raw_bytes"])
            return int((fetch + h["WRITE_SIZE_bytes"]) * B / frames), os.path.relpath(path, ROOT)
        except Exception:
            continue
    return None, None


def isa_entry(kernel_name, code=None, variant="min"):
    """instruction histogram of the kernel's iteration loop from the build's own assembly
    (ecc_ldpc_amd/build/isa_stats.json, written by build.py with tools/isa_histogram.py); for a run-time specialised
    kernel: from the assembly of its generated source, compiled once with the tool chain and kept next to the cached
    code object (jit_cache/<kernel>.isa.json)"""
    if kernel_name.startswith("ldpc_jit_") and code is not None:
        try:
            import importlib.util
            import ecc_ldpc_amd as E
            cache = os.path.join(E.lib().ldpc_jit_cache_dir().decode(), kernel_name + ".isa.json")
            if os.path.exists(cache):
                ent = json.load(open(cache))
            else:
                spec = importlib.util.spec_from_file_location("isa_histogram", os.path.join(ROOT, "tools", "isa_histogram.py"))
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                ent = mod.run_on_source(code.jit_source(variant), os.path.join(ROOT, "ecc_ldpc_amd", "csrc"), kernel_name)[0]
                json.dump(ent, open(cache, "w"))
            return None if any(lp.get("inner_loops") for lp in ent["loops"]) else ent
        except Exception:
            return None
    try:
        stats = json.load(open(os.path.join(ROOT, "ecc_ldpc_amd", "build", "isa_stats.json")))
    except Exception:
        return None
    hits = [r for r in stats if kernel_name and kernel_name in r["kernel"]]
    if len(hits) != 1 or any(lp.get("inner_loops") for lp in hits[0]["loops"]):
        return None     # data-dependent inner loops (row-by-row generic kernel): a static count does not price a turn
    return hits[0]


def rooflines(args, dec, B, Eg, N, n_tx, k, s_bytes, launches, kernel_ms, turns_timed, live=False):
    """-> (roofline, second object or None).
    On-chip (fused) kernels keep the BP state in LDS/registers: HBM sees the LLRs in and the bits out, so the bound is
    VALU issue.  achieved = VALU wave-instructions per second = (VALU instructions one wave issues per ordinary turn of
    the iteration loop, counted in the build's assembly) x waves per frame x frame-turns REALLY run in the timed steps
    (sum of the per-frame iteration counts -- early exit priced as it happened) / summed launch time (HIP events on the
    launch stream); peak = 1024 SIMDs x 2.4 GHz / 2 clk (full-rate issue, MI355X_MICROARCH.md).  The contract's HBM
    byte model (a two-kernel decoder with state in HBM) is kept as `roofline_hbm_model`, priced with
    the same iteration sum; its frac > 1 is the on-chip residency, not a measurement of HBM."""
    avg_ms = kernel_ms / max(launches, 1)
    steps = max(args.steps, 1)
    frames_timed = steps * B
    B_iter = (3 * Eg + 3 * N) * s_bytes
    # sum over frames of iters_f * B_iter  +  per frame: one more syndrome pass (reads lam), LLRs in, bits out
    model_bytes = turns_timed * B_iter + frames_timed * (N * s_bytes + n_tx * s_bytes + (k + 7) // 8)
    hbm = {"bound": "hbm", "achieved": round(model_bytes / (kernel_ms * 1e-3) / 1e9, 1) if kernel_ms else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "bytes_model": "sum_frames(iters_f)*(3E+3N)*s + frames*(N*s + n_tx*s + ceil(k/8)) -- SURVEY.md section 8d priced with the iterations run",
           "algorithmic_bytes_timed": model_bytes, "mean_iters_timed": round(turns_timed / frames_timed, 3)}
    hbm["frac"] = round(hbm["achieved"] / HBM_PEAK_GBS, 4)
    traffic, tsrc = live_traffic(args, dec, B) if live else (None, None)
    if traffic is None:
        traffic, tsrc = committed_traffic(args, dec, B)
        tsrc = tsrc
    if dec.path != "fused" and dec.schedule == "layered":
        # layered schedule from HBM: ONE launch is the whole decode of the batch; per sweep every edge reads and writes its
        # lam cell and its message (4E*s), plus the syndrome pass before the first sweep (E*s); priced with the sweeps run
        records = dec.kernel_name.endswith(", true>")   # min-sum rows as {c1, c2, meta} records: 2*s + 4 bytes per ROW instead of s per edge
        on_chip_lam = "layered_lds_kernel" in dec.kernel_name   # r04: lam in LDS (fp16), only the 12-byte row records stream (layered_lds.hip)
        if on_chip_lam:
            llr_bytes = s_bytes                      # (dtype f16: the frame source writes fp16 LLRs)
            per_sweep = 24 * dec.code.M
            # the first sweep of a frame writes its records without reading any; LLRs read once (twice by a frame out of sweeps), N result bytes
            bytes_timed = turns_timed * per_sweep - frames_timed * 12 * dec.code.M + frames_timed * (N * llr_bytes + N)
        else:
            per_sweep = 2 * Eg * s_bytes + (2 * dec.code.M * (2 * s_bytes + 4) if records else 2 * Eg * s_bytes)
            bytes_timed = turns_timed * per_sweep + frames_timed * Eg * s_bytes
        ach = bytes_timed / (kernel_ms * 1e-3) / 1e9 if kernel_ms else 0.0
        r = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
             "traffic": traffic, "traffic_source": tsrc,
             "kernel": dec.kernel_name, "launches": launches, "avg_launch_ms": round(avg_ms, 4),
             "algorithmic_bytes_timed": bytes_timed, "algorithmic_bytes_per_launch": bytes_timed // steps, "frame_sweeps_timed": turns_timed,
             "bytes_model": ("sum_frames(sweeps_f)*24M - frames*12M + frames*(N*llr_bytes + N): one 12-byte record per row read+written per sweep (lam stays in LDS as fp16), LLRs in, bits out"
                             if on_chip_lam else "sum_frames(sweeps_f)*(2E*s + 2M*(2s+4)) + frames*E*s: lam cells read+written per edge, one record per row read+written"
                             if records else "sum_frames(sweeps_f)*4E*s + frames*E*s: lam cells and messages read+written per edge") +
                            " (one launch decodes the batch; state in HBM)"}
        return r, None
    if dec.path != "fused" and "flood_qc_kernel" in dec.kernel_name:
        # flooding from HBM, one workgroup per frame, ONE launch per batch: the contract's byte model is exactly this
        # kernel's algorithmic traffic -- (3E+3N)*s per frame and turn run, the LLRs in, the bits out
        r = dict(hbm)
        r.update({"traffic": traffic, "traffic_source": tsrc, "kernel": dec.kernel_name,
                  "launches": launches, "avg_launch_ms": round(avg_ms, 4), "frame_turns_timed": turns_timed,
                  "algorithmic_bytes_per_launch": model_bytes // steps})
        return r, None
    if dec.path != "fused":
        # flood path: state in HBM, two kernels per turn; the timed kernel is the check-node kernel and every launch of
        # it streams (2E+N)*s bytes per frame of the batch
        bytes_per_launch = B * (2 * Eg + N) * s_bytes
        ach = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if launches else 0.0
        r = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
             "traffic": None, "kernel": dec.kernel_name, "launches": launches, "avg_launch_ms": round(avg_ms, 4),
             "algorithmic_bytes_per_launch": bytes_per_launch, "bytes_model": "(2E+N)*s * frames per check-node launch"}
        return r, hbm
    threads, fpw = dec.kernel_geometry
    ent = isa_entry(dec.kernel_name, dec.code, "min" if args.variant == "minsum" else "tanh")
    r = {"bound": "valu", "unit": "G wave-instr/s", "peak": round(VALU_PEAK / 1e9, 1), "kernel": dec.kernel_name, "launches": launches,
         "avg_launch_ms": round(avg_ms, 4), "traffic": traffic, "traffic_source": tsrc,
         "frame_turns_timed": turns_timed, "threads_per_workgroup": threads, "frames_per_workgroup": fpw}
    busy = live_pipe_busy(args, dec, B) if live else None
    if busy:
        r["counters"] = busy
    if ent is None or not threads or not fpw or not kernel_ms:
        r.update({"achieved": None, "frac": None, "note": "no static instruction count for this kernel instance (build/isa_stats.json)"})
        return r, hbm
    loops = ent["loops"]          # one per wave-group program; a wave runs exactly one of them, groups are equal in size
    valu_wave = sum(lp["hot_turn"]["units"].get("valu", 0) for lp in loops) / len(loops)
    clk_wave = sum(lp["hot_turn"]["valu_cost_weighted_clk"] for lp in loops) / len(loops)
    lds_wave = sum(lp["hot_turn"]["units"].get("lds", 0) for lp in loops) / len(loops)
    waves_per_frame = threads / 64.0 / fpw
    t = kernel_ms * 1e-3
    ach = valu_wave * waves_per_frame * turns_timed / t
    by_class = {}
    for lp in loops:
        for c, v in lp["hot_turn"]["valu_by_class"].items():
            by_class[c] = by_class.get(c, 0) + v / len(loops)
    r.update({"achieved": round(ach / 1e9, 1), "frac": round(ach / VALU_PEAK, 4),
              "valu_instr_per_wave_turn": round(valu_wave, 1), "valu_instr_per_wave_turn_by_issue_class": {c: round(v, 1) for c, v in sorted(by_class.items())},
              "lds_instr_per_wave_turn": round(lds_wave, 1), "waves_per_frame": waves_per_frame,
              # share of all SIMD cycles in which the VALU pipe was busy if every instruction took exactly its issue cost
              "valu_pipe_busy_frac": round(clk_wave * waves_per_frame * turns_timed / (N_SIMD * CLOCK_HZ * t), 4),
              "isa_source": "ecc_ldpc_amd/build/isa_stats.json = tools/isa_histogram.py over this build's own assembly (hot turn of the iteration loop; "
                            "prologue, the final syndrome-only pass and convergence snapshots are not counted: achieved is a lower bound)"})
    return r, hbm


def proof_of_work(args, E, ecc, dec, sim, llr_t, msg_t, bits, iters_t, conv_t, sp, f16):
    """Outside the timed region: the kernel that was timed against the flood path (a different implementation: two
    kernels per turn, state in HBM, one codeword per lane) on a 1024-frame sample, at the metric's Eb/N0 (below the
    waterfall nearly every frame fails, and a failed frame outputs the channel's hard decisions by the reference's
    rule Orig.hs:70 -- so agreement there shows little) AND at a point in the waterfall, where the frames converge
    at different turns: identical hard bits, iteration counts and flags, CRC32 of each printed."""
    import zlib
    import torch
    n = min(1024, llr_t.shape[0])
    k = ecc.message_length
    if dec.schedule == "layered" or args.dtype == "f16pk":   # one implementation each: checked by invariant, judged by another kernel
        return proof_of_work_layered(args, E, ecc, dec, sim, llr_t, msg_t, bits, iters_t, conv_t, sp, f16, n)
    flood_dtype = "f32" if args.dtype == "f16" else args.dtype     # fused-F16(llr) == F32 decoder on the fp16-rounded LLRs
    variant = "min" if args.variant == "minsum" else "tanh"
    other = "fused" if dec.path == "flood" else "flood"     # a different implementation of the same schedule
    out = {"sample_frames": n, "checked_against": f"{other} path (another kernel family for the same decoder)", "points": []}
    try:
        flood = E.Decoder(ecc.code, variant, flood_dtype, n, path=other)
    except E.LdpcError as e:
        out["error"] = str(e)
        return out
    fb = torch.empty_like(bits[:n])
    fi = torch.empty_like(iters_t[:n])
    fc = torch.empty_like(conv_t[:n])
    for db in (args.ebn0, args.ebn0 + 1.6):
        sim.generate(args.seed, 1 << 40, n, db, llr_t.data_ptr(), msg_t.data_ptr(), sp, llr_f16=f16)
        dec.decode_batch_dev(llr_t.data_ptr(), bits.data_ptr(), n, args.iters, iters_t.data_ptr(), conv_t.data_ptr(), sp, llr_f16=f16)
        flood.decode_batch_dev(llr_t.data_ptr(), fb.data_ptr(), n, args.iters, fi.data_ptr(), fc.data_ptr(), sp, llr_f16=f16)
        torch.cuda.synchronize()
        b, it, cv = bits[:n].cpu().numpy(), iters_t[:n].cpu().numpy(), conv_t[:n].cpu().numpy()
        raw = ((llr_t[:n, :k] > 0).to(torch.uint8) != msg_t[:n]).sum().item()
        dec_err = (bits[:n, :k] != msg_t[:n]).sum().item()
        out["points"].append({
            "ebn0_db": db, "bits_equal": bool((bits[:n] == fb).all().item()), "iters_equal": bool((iters_t[:n] == fi).all().item()),
            "converged_equal": bool((conv_t[:n] == fc).all().item()), "crc32_bits": zlib.crc32(b.tobytes()), "crc32_iters": zlib.crc32(it.tobytes()),
            "converged_frac": round(float(cv.mean()), 4), "mean_iters": round(float(it.mean()), 2), "distinct_iteration_counts": int(len(set(it.tolist()))),
            "channel_bit_errors": int(raw), "decoded_bit_errors": int(dec_err)})
    flood.close()
    out["ok"] = all(p["bits_equal"] and p["iters_equal"] and p["converged_equal"] for p in out["points"])
    return out


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
    # chunks of frames until the time budget is spent (a single probe mis-predicts: short bursts run faster than the
    # box's sustained CPU quota allows)
    chunk = max(cores, min(llr_dev.shape[0], 4 * cores))
    n, dt = 0, 0.0
    while n + chunk <= llr_dev.shape[0] and dt < args.cpu_seconds:
        x = llr_dev[n:n + chunk].cpu().numpy().astype(np.float64)
        t0 = time.perf_counter()
        if args.schedule == "layered":
            oracle.decode_layered_batch(g, code.layers(), variant, args.iters, x, nthreads=cores)
        else:
            oracle.decode_batch(g, variant, args.iters, x, nthreads=cores)
        dt += time.perf_counter() - t0
        n += chunk
    v = n * ecc.message_length / dt / 1e6
    return {"value": round(v, 4), "unit": "Mbit/s", "cores": cores, "kind": "port",
            "sample": f"first {n} frames of the step-0 batch, {args.iters} iters, double precision, {cores} threads, {dt:.1f} s",
            "gpu_over_cpu": round(gpu_value / v, 1) if v > 0 else None}


if __name__ == "__main__":
    main()  # normal interpreter exit: rocprofv3 writes its output from exit handlers
