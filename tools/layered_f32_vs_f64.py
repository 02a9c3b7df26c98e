#!/usr/bin/env python3
"""tools/layered_f32_vs_f64.py -- the layered f32 kernels against the Double layered specification (oracle_decode_layered), over
>= 10 000 frames across the waterfall: how many converged flags flip, how the sweep counts differ, whether the hard bits of the
frames both decode are equal.  (The serial schedule amplifies a rounding difference faster than flooding does: a frame at the edge of
convergence may fall the other way.  tests/test_layered_fused_gpu.py and tests/test_layered_gpu.py take their bars from this output:
profiles/r04_layered_f32_vs_f64.txt.)"""
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
    tot = Counter()
    for name, frames, dbs in (("jpl.1024.4.5", 2000, (2.5, 3.0, 3.5)), ("jpl.4096.4.5", 1400, (2.6, 2.9, 3.2))):
        c = load(name)
        lp = np.arange(0, c.M + 1, c.sz, dtype=np.int32)
        for variant in ("min", "tanh"):
            for db in dbs:
                _, llr = c.frames(frames, db, seed=int(db * 100) + 9400)
                outs = {}
                for path in (("auto",) if variant == "tanh" else ("auto", "flood")):     # (tanh: the HBM kernel only; min-sum: on-chip and HBM)
                    dec = E.Decoder(c.hip_code(E), variant, "f32", frames, schedule="layered", path=path)
                    outs[path] = (dec.kernel_name, dec.decode_batch(llr.astype(np.float32), 50))
                    dec.close()
                ob, oi, oc = oracle.decode_layered_batch(c.graph, lp, variant, 50, llr, nthreads=cores)
                oc = oc.astype(bool)
                for path, (kname, (bits, its, conv)) in outs.items():
                    conv = conv.astype(bool)
                    same = conv == oc
                    d = (its.astype(int) - oi.astype(int))[same]
                    hist = dict(sorted(Counter(d.tolist()).items()))
                    big = {k: v for k, v in hist.items() if abs(k) > 1}
                    print(f"{name:13s} {variant:4s} {db:3.1f} dB {kname[:40]:40s} frames {frames} oracle converged {int(oc.sum()):5d}  flags flipped {int((~same).sum()):3d} "
                          f"(f32 only {int((conv & ~oc).sum())}, f64 only {int((oc & ~conv).sum())})  bits equal where flags agree {np.array_equal(bits[same], ob[same])}  "
                          f"sweeps equal {np.mean(d == 0):.4f} within one {np.mean(np.abs(d) <= 1):.4f}  beyond one: {big}", flush=True)
                    tot[(variant, "frames")] += frames; tot[(variant, "flips")] += int((~same).sum()); tot[(variant, "sweeps_ne")] += int((d != 0).sum())
                    tot[(variant, "sweeps_gt1")] += int((np.abs(d) > 1).sum()); tot[(variant, "bits_ne")] += int((bits[same] != ob[same]).any(axis=1).sum())
                if "flood" in outs:
                    a, b = outs["auto"][1], outs["flood"][1]
                    assert all(np.array_equal(x, y) for x, y in zip(a, b)), "on-chip and HBM layered kernels differ"
    for v in ("min", "tanh"):
        n = tot[(v, "frames")]
        print(f"TOTAL {v}: {n} frame decodes, flags flipped {tot[(v, 'flips')]} ({tot[(v, 'flips')] / n:.2e}), sweep counts differing {tot[(v, 'sweeps_ne')]} ({tot[(v, 'sweeps_ne')] / n:.2e}), "
              f"by more than one {tot[(v, 'sweeps_gt1')]}, frames with unequal bits among agreeing flags {tot[(v, 'bits_ne')]}")


if __name__ == "__main__":
    main()
