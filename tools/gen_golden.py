#!/usr/bin/env python3
"""Generates tests/golden/*.npz: known-answer vectors for the decode path.

The reference (ku-fpg/ecc-ldpc) ships no golden vectors and is Haskell (cannot run here), so these
are produced by the build's own restatements and cross-checked before being written:
  * oracle/literal.py   (line-by-line Python transliteration of Reference/Orig.hs, Reference/Min.hs)
  * oracle/ldpc_oracle.c dense and sparse forms
All three must agree BIT FOR BIT on every stored value, otherwise this script aborts.
Run from the repository root:  python tools/gen_golden.py
A fixture holds data only: H (CSR), input LLRs (float32-exact), lam at the top of every loop turn,
the messages ne' of every update, final bits, iteration count, converged flag."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import literal, oracle  # noqa: E402
from tests.helpers import load  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
CASES = [
    # name, variant, max_iters, [(ebn0, seed)], use_literal_python
    ("moon.7.13", "tanh", 20, [(1.0, 11), (3.0, 12), (5.0, 13), (0.0, 14)], True),
    ("moon.7.13", "min", 20, [(1.0, 21), (3.0, 22), (5.0, 23), (0.0, 24)], True),
    ("jpl.1024.4.5", "tanh", 12, [(3.0, 31), (4.0, 32), (2.0, 33)], False),
    ("jpl.1024.4.5", "min", 12, [(3.0, 41), (4.0, 42), (2.0, 43)], False),
    ("1920.1280.3.303", "tanh", 10, [(2.5, 51), (4.0, 52)], False),
    ("1920.1280.3.303", "min", 10, [(2.5, 61), (4.0, 62)], False),
    # the headline code (added in round 2; these fixtures also carry the QC description so that the GPU test can run the
    # fused QC kernels on them, not only the CSR paths)
    ("jpl.4096.4.5", "min", 10, [(3.4, 71), (3.9, 72), (2.0, 73)], False),
    ("jpl.4096.4.5", "tanh", 8, [(3.4, 81), (2.0, 82)], False),
    # the redundant-check matrix of the same code as 1920.1280.3.303 (added in round 3): 5760 x 1920, E = 32 000, column weight 18
    ("1920.1280.A", "tanh", 5, [(1.5, 91), (3.0, 92)], False),
    ("1920.1280.A", "min", 6, [(1.5, 93), (4.5, 94)], False),
]


def main():
    os.makedirs(OUT, exist_ok=True)
    only = [a for a in sys.argv[1:] if not a.startswith("-")]   # e.g. `python tools/gen_golden.py jpl.4096.4.5`
    for name, variant, iters, frames, use_literal in CASES:
        if only and name not in only:
            continue
        c = load(name)
        rec = dict(row_ptr=c.graph.row_ptr, col_idx=c.graph.col_idx, N=np.int32(c.N), max_iters=np.int32(iters),
                   variant=np.array(variant))
        if name == "jpl.4096.4.5":
            rec["qc_sz"] = np.int32(c.sz)
            rec["qc_offsets"] = c.offsets.astype(np.int32)
        for i, (db, seed) in enumerate(frames):
            cw, llr = c.frames(1, db, seed)
            llr = llr[0]
            o = oracle.decode(c.graph, variant, iters, llr, trace=True)
            d = oracle.decode_dense(c.H, variant, iters, llr, trace=True)
            assert o["iters"] == d["iters"] and np.array_equal(o["trace_lam"], d["trace_lam"]) and np.array_equal(o["bits"], d["bits"])
            if use_literal or True:  # literal Python on every case (jpl.1024 costs a few seconds per frame)
                tr = []
                b, it, cv = literal.ldpc(c.H, variant, iters, llr, trace=tr)
                assert it == o["iters"] and cv == o["converged"] and np.array_equal(b, o["bits"])
                assert np.array_equal(np.array(tr), o["trace_lam"]), (name, variant, db)
            rec[f"f{i}_ebn0"] = np.float64(db)
            rec[f"f{i}_codeword"] = cw[0]
            rec[f"f{i}_llr"] = llr.astype(np.float32)
            rec[f"f{i}_trace_lam"] = o["trace_lam"]
            rec[f"f{i}_trace_ne"] = o["trace_ne"]
            rec[f"f{i}_bits"] = o["bits"]
            rec[f"f{i}_iters"] = np.int32(o["iters"])
            rec[f"f{i}_converged"] = np.bool_(o["converged"])
            print(f"{name} {variant} {db} dB: iters {o['iters']} converged {o['converged']}", flush=True)
        rec["n_frames"] = np.int32(len(frames))
        path = os.path.join(OUT, f"{name}.{variant}.npz")
        np.savez_compressed(path, **rec)
        print("wrote", path, os.path.getsize(path), "bytes", flush=True)


if __name__ == "__main__":
    main()
