#!/usr/bin/env python3
"""Throughput of the run-time specialised QC kernels next to the built-in instance of the headline code (GPU box).
Frames: all-zero codeword + AWGN generated with torch on the device (any linear code), decoded device-resident.
  python tools/jit_bench.py [--frames 65536] [--iters 50] [--ebn0 2.0]
Prints one row per code/variant: kernel, waves, ms per launch, Mbit/s, and VALU-turn rate = frames * iters_run * E / time
(edge-updates per second: comparable across codes, unlike Mbit/s)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import ecc_ldpc_amd as E
from tests.helpers import SYNTHETIC_NAMES, load, synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=65536)
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--ebn0", type=float, default=2.0)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--codes", default="jpl.4096.4.5," + ",".join(SYNTHETIC_NAMES))
args = ap.parse_args()
E.init(0)
dev = torch.device("cuda", 0)
for name in args.codes.split(","):
    c = load(name) if name.startswith("jpl.") else synthetic(name)
    code = c.hip_code(E)
    k, n_tx = c.k, c.n_tx
    F = min(args.frames, max(1024, int(args.frames * 5632 / c.N)))
    s2 = 1.0 / (2.0 * (k / n_tx) * 10 ** (args.ebn0 / 10))
    g = torch.Generator(device=dev); g.manual_seed(1)
    llr = torch.zeros((F, c.N), dtype=torch.float32, device=dev)
    llr[:, :n_tx] = (2.0 / s2) * (-1.0 + torch.randn((F, n_tx), generator=g, device=dev) * s2 ** 0.5)
    bits = torch.empty((F, c.N), dtype=torch.uint8, device=dev)
    iters = torch.empty((F,), dtype=torch.int32, device=dev)
    for variant in ("min", "tanh"):
        t0 = time.time()
        dec = E.Decoder(code, variant, "f32", F)
        t_create = time.time() - t0
        st = torch.cuda.Stream(device=dev)
        dec.decode_batch_dev(llr.data_ptr(), bits.data_ptr(), F, args.iters, iters.data_ptr(), None, st.cuda_stream)
        torch.cuda.synchronize()
        dec.set_timing(True)
        for _ in range(args.steps):
            dec.decode_batch_dev(llr.data_ptr(), bits.data_ptr(), F, args.iters, iters.data_ptr(), None, st.cuda_stream)
        torch.cuda.synchronize()
        n, ms = dec.kernel_time()
        turns = int(iters.sum().item())
        ms /= n
        print(f"{name:22s} {variant:4s} {dec.kernel_name[:46]:46s} thr/wg {dec.kernel_geometry[0]:4d} frames {F:6d} {ms:8.3f} ms  "
              f"{F * k / ms / 1e3:9.1f} Mbit/s  {turns * c.E / ms / 1e9:7.2f} T edge-updates/s  mean iters {turns / F:5.1f}  (create {t_create:.1f} s)", flush=True)
        dec.close()
