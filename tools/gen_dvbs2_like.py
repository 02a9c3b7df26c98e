#!/usr/bin/env python3
"""Writes codes/dvbs2like.64800.1.2/H.q: a SYNTHETIC parity-check matrix with the SHAPE of the DVB-S2 rate-1/2
normal-frame code -- n = 64 800, k = 32 400, period 360 (circulant size), q = 90 block rows, an irregular
repeat-accumulate structure: information part with 36 block columns of weight 8 and 54 of weight 3 (every check sees
5 information circulants), parity part dual-diagonal -- for BASELINE.json configs[4] ("DVB-S2 n=64800 long code,
layered min-sum + early termination").  The ETSI EN 302 307 address tables are NOT in the reference repository or in
this container, so the rotations and the placement of the information circulants are pseudo-random (fixed seed, at
most one circulant per block, 4-cycles between information circulants avoided): same size, degree profile and memory
behaviour as the real code, NOT its BER.  It is not the standard's matrix and must not be used as such.
No generator is written (a dense G would be 131 MB); frames are the all-zero codeword.
r04: the block rows are written in an order in which runs of FOUR consecutive rows share no block column (group_order below), the
way layered hardware decoders schedule such codes: a layered decoder visits the block rows in the order of the file, and block rows
that touch distinct columns can be worked on together without changing any result (csrc/layered_lds.hip does).  Same matrix up to a
permutation of its rows, same code.
  python tools/gen_dvbs2_like.py [--check]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SZ, Q, KB = 360, 90, 90           # circulant size, block rows, information block columns (parity block columns: Q)


def build(seed=0x0D5B2):
    rng = np.random.default_rng(seed)
    off = -np.ones((Q, KB + Q), np.int64)
    colw = [8] * 36 + [3] * 54                          # information block-column weights (sum = 450 = 5 per block row)
    # deal the column stubs to the block rows so that every block row gets exactly 5 and no block is hit twice
    for _ in range(1000):
        stubs = [bc for bc, w in enumerate(colw) for _ in range(w)]
        rng.shuffle(stubs)
        rows = [stubs[i * 5:(i + 1) * 5] for i in range(Q)]
        bad = [i for i, r in enumerate(rows) if len(set(r)) < 5]
        tries = 0
        while bad and tries < 100000:                   # swap a duplicated stub with a stub of another row
            i = bad[0]
            r = rows[i]
            dup = next(k for k in range(5) if r.count(r[k]) > 1)
            j = int(rng.integers(0, Q)); kk = int(rng.integers(0, 5))
            if j != i and rows[j][kk] not in r and r[dup] not in rows[j]:
                r[dup], rows[j][kk] = rows[j][kk], r[dup]
            bad = [x for x, rr in enumerate(rows) if len(set(rr)) < 5]
            tries += 1
        if not bad:
            break
    assert not bad
    for br, r in enumerate(rows):
        for bc in r:
            off[br, bc] = rng.integers(0, SZ)
    # remove 4-cycles between information circulants: rows a, b and columns c, d with off[a,c]-off[a,d]-off[b,c]+off[b,d] = 0 (mod SZ)
    for _ in range(50):
        fixed = 0
        cols_of = [np.nonzero(off[br, :KB] >= 0)[0] for br in range(Q)]
        for a in range(Q):
            for b in range(a + 1, Q):
                common = np.intersect1d(cols_of[a], cols_of[b])
                for i in range(len(common)):
                    for j in range(i + 1, len(common)):
                        c, d = common[i], common[j]
                        if (off[a, c] - off[a, d] - off[b, c] + off[b, d]) % SZ == 0:
                            off[b, d] = (off[b, d] + int(rng.integers(1, SZ))) % SZ
                            fixed += 1
        if not fixed:
            break
    # accumulator: T[i][i] = I, T[i][i-1] = I, and the wrap block T[0][Q-1] a shift by one
    for i in range(Q):
        off[i, KB + i] = 0
        if i:
            off[i, KB + i - 1] = 0
    off[0, KB + Q - 1] = 1
    return off.astype(np.int32)


def group_order(off, gmax=4):
    """permutation of the block rows into runs of up to `gmax` consecutive rows that pairwise share no block column (greedy: start a
    run with the row that has the fewest compatible rows left, extend it with the row compatible with all of the run that has the
    fewest compatible rows left); full runs first, shorter ones after them"""
    Q = off.shape[0]
    hit = off >= 0
    ok = ~((hit[:, None, :] & hit[None, :, :]).any(axis=2))
    np.fill_diagonal(ok, False)
    free = set(range(Q))
    runs = []
    while free:
        deg = {i: sum(1 for j in free if ok[i, j]) for i in free}
        run = [min(free, key=lambda i: (deg[i], i))]
        free.discard(run[0])
        while len(run) < gmax:
            cand = [j for j in free if all(ok[i, j] for i in run)]
            if not cand:
                break
            j = min(cand, key=lambda j: (deg[j], j))
            run.append(j); free.discard(j)
        runs.append(sorted(run))
    runs.sort(key=lambda r: (-len(r), r))
    return np.array([r for run in runs for r in run]), [len(r) for r in runs]


def main():
    off = build()
    order, sizes = group_order(off)
    off = off[order]
    print(f"block rows reordered into runs of independent rows: sizes {sorted(set(sizes), reverse=True)} x {[sizes.count(z) for z in sorted(set(sizes), reverse=True)]}")
    out = os.path.join(ROOT, "codes", "dvbs2like.64800.1.2")
    os.makedirs(out, exist_ok=True)
    lines = [str(SZ)]
    for br in range(off.shape[0]):
        lines.append(" ".join(str(1 << int(v)) if v >= 0 else "0" for v in off[br]))    # QuasiCyclic.hs:52-56: bit `off` of the first row
    open(os.path.join(out, "H.q"), "w").write("\n".join(lines) + "\n")
    open(os.path.join(out, "README"), "w").write(
        "SYNTHETIC matrix with the shape of the DVB-S2 rate-1/2 normal-frame LDPC code (n = 64800, k = 32400, period 360).\n"
        "Generated by tools/gen_dvbs2_like.py (fixed seed); NOT the ETSI EN 302 307 matrix -- the standard's address tables are\n"
        "not available here.  No generator matrix: frames are the all-zero codeword.\n"
        "Block rows are written in an order in which runs of four consecutive rows (4i .. 4i+3) share no block column (r04): a layered\n"
        "decoder visits them in file order and may work on such a run at once with unchanged results (csrc/layered_lds.hip).\n")
    rw = (off >= 0).sum(1)
    cw = (off >= 0).sum(0)
    print(f"wrote {out}/H.q: {off.shape[0]} x {off.shape[1]} blocks of {SZ}: N = {off.shape[1] * SZ}, M = {off.shape[0] * SZ}, "
          f"E = {int((off >= 0).sum()) * SZ}, row weights {sorted(set(rw.tolist()))}, column weights {sorted(set(cw.tolist()))}")


if __name__ == "__main__":
    main()
