#!/usr/bin/env python3
"""Differential fuzz of the on-chip QC kernels (built-in / run-time specialised / table-driven / generic) against the HBM
flood path on random single-circulant protographs: circulant sizes that are and are not powers of two, 2..16 block rows,
row weights 2..24, punctured-looking weight-1 columns included.  Both paths implement the same decoder; for f32 they must
agree bit for bit (hard bits, iteration counts, converged flags); so must the kernels of the row-layered schedule (two from HBM,
r03: one on-chip), and the packed-fp16 kernels (r03) reproduce their emulation bit for bit.  Usage: python tools/fuzz_qc.py [n_codes] [first_seed] [r04 = only the kinds added in round 4]
Prints one line per (code, rule); exit status 1 on the first disagreement."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ecc_ldpc_amd as E  # noqa: E402
from tests.helpers import SyntheticQC  # noqa: E402

SIZES = [16, 24, 27, 32, 48, 54, 64, 81, 96, 100, 128, 160, 192, 256, 360, 384]


def random_code(seed):
    rng = np.random.default_rng(seed)
    sz = int(rng.choice(SIZES))
    R = int(rng.integers(2, 17))
    C = int(rng.integers(R + 1, min(4 * R, 48) + 1))
    while C * sz * 4 > 150 * 1024:
        C -= 1
    if C <= R:
        return None
    mask = np.zeros((R, C), bool)
    wmax = min(C, 24)
    for br in range(R):
        w = int(rng.integers(2, max(3, min(wmax, 3 + C // 2)) + 1))
        mask[br, rng.choice(C, w, replace=False)] = True
    for bc in range(C):                       # no empty column; most columns weight >= 2
        need = 1 if rng.random() < 0.15 else 2
        while mask[:, bc].sum() < min(need, R):
            mask[int(rng.integers(0, R)), bc] = True
    off = np.where(mask, rng.integers(0, sz, mask.shape), -1).astype(np.int32)
    return SyntheticQC(f"fuzz{seed}-{R}x{C}-sz{sz}", sz, off)


def fuzz_r03_kinds(c, code, llr, hbm_layered, hbm_result):
    """r03: the on-chip layered kernel (f32: bit for bit the HBM layered kernel) and the two packed-fp16 kernels (bit for bit the
    emulation oracle/emulate_f16.py, on the first frames -- the emulation is numpy) specialised for this code at run time"""
    from oracle import emulate_f16 as em
    bad = 0
    F = len(llr)
    try:
        on = E.Decoder(code, "min", "f32", F, schedule="layered", path="fused")
    except E.LdpcError as e:
        print(f"{'':26s} min  no on-chip layered kernel ({str(e)[:60]}...)", flush=True)
        return 0
    a = on.decode_batch(llr, 20)
    same = all(np.array_equal(x, y) for x, y in zip(a, hbm_result))
    print(f"{'':26s} min  layered {on.kernel_name[:50]:50s} vs {hbm_layered.kernel_name[:20]:20s} {'ok' if same else 'MISMATCH'}", flush=True)
    bad += 0 if same else 1
    nf = 10
    sub = np.concatenate([llr[: nf // 2], llr[-nf // 2:]])        # both Eb/N0 halves
    for sched, fn in (("flooding", em.decode_minsum_pk16), ("layered", em.decode_minsum_pk16_layered)):
        try:
            pk = E.Decoder(code, "min", "f16pk", F, schedule=sched)
        except E.LdpcError as e:
            print(f"{'':26s} min  f16pk {sched}: no kernel ({str(e)[:60]}...)", flush=True)
            continue
        got = pk.decode_batch(sub, 30)
        eb, ei, ec, et = fn(c.graph, sub, 30)
        same = np.array_equal(got[0], eb) and np.array_equal(got[1], ei) and np.array_equal(got[2].astype(bool), ec)
        finite = all(np.isfinite(t).all() for t in et)
        print(f"{'':26s} min  f16pk {sched:8s} {pk.kernel_name[:44]:44s} vs emulation: converged {ec.mean():.2f} mean iters {ei.mean():5.1f} "
              f"{'ok' if same and finite else 'MISMATCH'}", flush=True)
        bad += 0 if same and finite else 1
    return bad


def fuzz_r04_kinds(c, llr):
    """r04: the code given as a plain CSR graph -- the generic on-chip kernel (its STAGED instance: f32 LLRs travel into LDS a frame
    ahead) against the batch-major HBM pair, bit for bit; layered min-sum with fp16 lam on-chip and streamed row records against
    its emulation (oracle/emulate_f16.py decode_minsum_f16_layered) on the first and last frames"""
    from oracle import emulate_f16 as em
    bad = 0
    F = len(llr)
    plain = E.Code.from_csr(c.graph.row_ptr, c.graph.col_idx, c.N)
    for rule in ("min", "tanh"):
        try:
            on = E.Decoder(plain, rule, "f32", F, path="fused")
        except E.LdpcError as e:
            print(f"{'':26s} {rule:4s} as CSR: no on-chip kernel ({str(e)[:50]}...)", flush=True)
            continue
        fl = E.Decoder(plain, rule, "f32", F, path="flood")
        a, b = on.decode_batch(llr, 30), fl.decode_batch(llr, 30)
        same = all(np.array_equal(x, y) for x, y in zip(a, b))
        print(f"{'':26s} {rule:4s} as CSR  {on.kernel_name[:66]:66s} vs {fl.kernel_name[:16]:16s} {'ok' if same else 'MISMATCH'}", flush=True)
        bad += 0 if same else 1
        del on, fl
    try:
        lds = E.Decoder(c.hip_code(E), "min", "f16", F, schedule="layered", path="flood")
    except E.LdpcError as e:
        print(f"{'':26s} min  layered f16: no kernel ({str(e)[:60]}...)", flush=True)
        return bad
    sub = np.concatenate([llr[:5], llr[-5:]])
    got = lds.decode_batch(sub, 30)
    eb, ei, ec, _ = em.decode_minsum_f16_layered(c.graph, sub, 30)
    same = np.array_equal(got[0], eb) and np.array_equal(got[1], ei) and np.array_equal(got[2].astype(bool), ec)
    print(f"{'':26s} min  layered f16 {lds.kernel_name[:40]:40s} vs emulation: converged {ec.mean():.2f} mean sweeps {ei.mean():5.1f} {'ok' if same else 'MISMATCH'}", flush=True)
    return bad + (0 if same else 1)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    only_r04 = len(sys.argv) > 3 and sys.argv[3] == "r04"
    E.init(0)
    F, bad = 192, 0
    for seed in range(first, first + n):
        c = random_code(seed)
        if c is None:
            continue
        code = c.hip_code(E)
        llr = np.concatenate([c.frames(F // 2, 2.5, seed)[1], c.frames(F // 2, 6.0, seed + 1)[1]]).astype(np.float32)
        if only_r04:   # (the kernels of r04_kinds are built in: no run-time compilation, hundreds of codes in minutes)
            print(f"{c.name:26s} rows {c.offsets.shape[0]:2d} max row weight {int((c.offsets >= 0).sum(1).max()):2d}", flush=True)
            bad += fuzz_r04_kinds(c, llr)
            continue
        for rule in ("min", "tanh"):
            t0 = time.time()
            try:
                fused = E.Decoder(code, rule, "f32", F, path="fused")
            except E.LdpcError as e:      # a frame no on-chip kernel holds: LDPC_PATH_AUTO would take the HBM path
                print(f"{c.name:26s} {rule:4s} no on-chip kernel ({str(e)[:60]}...)", flush=True)
                fused = None
            flood = E.Decoder(code, rule, "f32", F, path="flood")
            b = flood.decode_batch(llr, 30)
            a = fused.decode_batch(llr, 30) if fused else b
            same = all(np.array_equal(x, y) for x, y in zip(a, b))
            if fused:
                print(f"{c.name:26s} {rule:4s} rows {c.offsets.shape[0]:2d} max row weight {int((c.offsets >= 0).sum(1).max()):2d} "
                      f"{fused.kernel_name[:44]:44s} vs {flood.kernel_name[:28]:28s} converged {a[2].mean():.2f} mean iters {a[1].mean():5.1f} "
                      f"{'ok' if same else 'MISMATCH'}  ({time.time() - t0:.0f} s)", flush=True)
            if not same:
                bad += 1
                d = [int((x != y).sum()) for x, y in zip(a, b)]
                print("   differing (bits, iters, conv):", d, flush=True)
            del fused, flood
            # row-layered schedule: the frame-per-workgroup QC kernel (min-sum: row records) against the batch-major any-H kernel
            qc = E.Decoder(code, rule, "f32", F, schedule="layered", path="flood")
            os.environ["LDPC_LAYERED_QC"] = "0"
            bm = E.Decoder(code, rule, "f32", F, schedule="layered", path="flood")
            del os.environ["LDPC_LAYERED_QC"]
            a = qc.decode_batch(llr, 20)
            b = bm.decode_batch(llr, 20)
            same = all(np.array_equal(x, y) for x, y in zip(a, b))
            print(f"{'':26s} {rule:4s} layered {qc.kernel_name[:50]:50s} vs {bm.kernel_name[:20]:20s} converged {a[2].mean():.2f} mean sweeps {a[1].mean():5.1f} "
                  f"{'ok' if same else 'MISMATCH'}", flush=True)
            if not same:
                bad += 1
                print("   differing (bits, sweeps, conv):", [int((x != y).sum()) for x, y in zip(a, b)], flush=True)
            del bm
            if rule == "min":
                bad += fuzz_r03_kinds(c, code, llr, qc, a)
            del qc
        bad += fuzz_r04_kinds(c, llr)
    print("fuzz:", "FAILED" if bad else "all equal")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
