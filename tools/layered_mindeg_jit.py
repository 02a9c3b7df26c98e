#!/usr/bin/env python3
"""Which rows should the on-chip layered kernels split between their two wave groups?  (fused_layered_body.h LAY_SPLIT_MIN_DEG: rows
lighter than the threshold stay whole with one group -- one barrier per layer instead of two.)  Run-time specialised kernels of
synthetic QC codes with two wave groups, compiled with the threshold given through LDPC_JIT_EXTRA_OPTS (never cached), timed on
device-resident frames.  GPU box:  python tools/layered_mindeg_jit.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import ecc_ldpc_amd as E
from tests.helpers import synthetic

E.init(0)
dev = torch.device("cuda", 0)
for name, ebn0 in (("irregular-20x30-sz64", 4.0), ("dvbs2short-20x45-sz360", 2.5), ("wide-4x40-sz256", 4.0)):
    c = synthetic(name)
    code = c.hip_code(E)
    F = max(2048, int(32768 * 5632 / c.N)) // 2 * 2
    s2 = 1.0 / (2.0 * (c.k / c.n_tx) * 10 ** (ebn0 / 10))
    g = torch.Generator(device=dev); g.manual_seed(1)
    llr = torch.zeros((F, c.N), dtype=torch.float32, device=dev)
    llr[:, :c.n_tx] = (2.0 / s2) * (-1.0 + torch.randn((F, c.n_tx), generator=g, device=dev) * s2 ** 0.5)
    bits = torch.empty((F, c.N), dtype=torch.uint8, device=dev)
    iters = torch.empty((F,), dtype=torch.int32, device=dev)
    degs = sorted(set((c.offsets >= 0).sum(1).tolist()))
    for dtype in ("f32", "f16pk"):
        row = []
        for md in (2, 4, 6, 8, 12, 33):
            os.environ["LDPC_JIT_EXTRA_OPTS"] = f"-DLAY_SPLIT_MIN_DEG={md}"
            os.environ["LDPC_JIT_NOCACHE"] = "1"
            dec = E.Decoder(code, "min", dtype, F, schedule="layered", path="fused")
            st = torch.cuda.Stream(device=dev)
            x = llr.half() if dtype == "f16pk" else llr
            run = lambda: dec.decode_batch_dev(x.data_ptr(), bits.data_ptr(), F, 50, iters.data_ptr(), None, st.cuda_stream, llr_f16=(dtype == "f16pk"))
            run(); torch.cuda.synchronize()
            dec.set_timing(True)
            for _ in range(4):
                run()
            torch.cuda.synchronize()
            n, ms = dec.kernel_time()
            row.append(f"{md}: {ms / n:7.3f} ms")
            dec.close()
        print(f"{name:24s} row weights {degs}  {dtype:5s} sweeps {iters.float().mean().item():5.1f}  LAY_SPLIT_MIN_DEG " + "  ".join(row), flush=True)
