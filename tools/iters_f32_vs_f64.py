#!/usr/bin/env python3
"""tools/iters_f32_vs_f64.py -- how often, and by how much, the f32 kernels' iteration counts differ from the Double oracle's.
Hard bits and converged flags are required identical by the parity tests; the iteration count of a frame can differ when a
syndrome bit flips one turn earlier or later because a float LLR and the Double LLR straddle zero.  Prints, per code / rule /
Eb/N0: frames, share with equal counts, histogram of (f32 - f64) turn differences, and whether bits / flags were equal."""
import os
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import ecc_ldpc_amd as E  # noqa: E402
from oracle import oracle  # noqa: E402
from tests.helpers import load  # noqa: E402


def main():
    E.init(0)
    cores = len(os.sched_getaffinity(0))
    for name, frames, dbs in (("jpl.1024.4.5", 1500, (2.5, 3.0, 3.5)), ("jpl.4096.4.5", 400, (2.8, 3.2)), ("1920.1280.3.303", 1500, (1.5, 2.5))):
        c = load(name)
        for variant in ("min", "tanh"):
            for db in dbs:
                _, llr = c.frames(frames, db, seed=int(db * 100) + 9000)
                dec = E.Decoder(c.hip_code(E), variant, "f32", frames)
                bits, its, conv = dec.decode_batch(llr.astype(np.float32), 50)
                ob, oi, oc = oracle.decode_batch(c.graph, variant, 50, llr, nthreads=cores)
                d = its.astype(int) - oi.astype(int)
                hist = dict(sorted(Counter(d.tolist()).items()))
                print(f"{name:16s} {variant:4s} {db:3.1f} dB  {dec.kernel_name[:34]:34s} frames {frames}  converged {int(oc.sum())}  bits equal {np.array_equal(bits, ob)}  "
                      f"flags equal {np.array_equal(conv, oc)}  iteration counts equal {np.mean(d == 0):.4f}  (f32 - f64) histogram {hist}", flush=True)
                dec.close()


if __name__ == "__main__":
    main()
