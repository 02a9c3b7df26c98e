"""NumPy restatement of the reference's matrix file formats.  TEST INFRASTRUCTURE ONLY
(see oracle/ldpc_oracle.c header: only tests/, smoke() and bench.py's cpu_baseline may import
anything under oracle/).  PARITY UNPINNED for the same reason as the decoder oracle: the
reference holds no fixture for its loaders; the pins are structural ([I|G] H^T = 0, H.alist ==
expand(H.q)), asserted in tests/test_formats.py.

Restated from (paths relative to /root/reference):
  src/Data/BitMatrix/Alist.hs:30-46        Read Alist: drop zeros, rows first, only row lists used
  src/Data/Matrix/QuasiCyclic.hs:19-25,52-56  .q = cycle size + Matlab matrix of integers; expansion
  src/Data/Matrix/Matlab.hs:20-22, src/Data/BitMatrix/Matlab.hs:20-26   .m = rows of 0/1 words
  src/ECC/Code/LDPC/Fast/Arraylet.hs:68-79 initMatrixlet: single set bit -> rotation (log2), else error
"""
from __future__ import annotations

import numpy as np


def read_alist_reference(text: str) -> np.ndarray:
    """Alist.hs:30-46.  All zero tokens are filtered out BEFORE parsing; first number = rows,
    second = cols; two ignored numbers; per-row counts; per-col counts; then the row lists
    (1-based column indices); the column lists are parsed and dropped."""
    toks = [int(t) for t in text.split()]
    toks = [t for t in toks if t != 0]
    pos = 0

    def item():
        nonlocal pos
        v = toks[pos]
        pos += 1
        return v

    n = item()
    m = item()
    item()
    item()
    num_n = [item() for _ in range(n)]
    num_m = [item() for _ in range(m)]
    H = np.zeros((n, m), dtype=np.uint8)
    for r, c in enumerate(num_n):
        for _ in range(c):
            H[r, item() - 1] = 1
    for c in num_m:
        for _ in range(c):
            item()
    return H


def read_alist_mackay(text: str) -> np.ndarray:
    """MacKay's published order (N M / max weights / column weights / row weights / column
    lists / row lists, zero padded).  The reference reader cannot load this (it would read the
    file transposed, SURVEY.md section 0 note ii); codes/1920.1280.3.303 is in this order.
    Returns H as M x N."""
    toks = [int(t) for t in text.split()]
    N, M = toks[0], toks[1]
    maxc, maxr = toks[2], toks[3]
    pos = 4
    colw = toks[pos:pos + N]
    pos += N
    roww = toks[pos:pos + M]
    pos += M
    H = np.zeros((M, N), dtype=np.uint8)
    for c in range(N):
        ent = toks[pos:pos + maxc]
        pos += maxc
        for r in ent[:colw[c]]:
            H[r - 1, c] = 1
    Hr = np.zeros_like(H)
    for r in range(M):
        ent = toks[pos:pos + maxr]
        pos += maxr
        for c in ent[:roww[r]]:
            Hr[r, c - 1] = 1
    if not np.array_equal(H, Hr):
        raise ValueError("MacKay alist: column lists and row lists disagree")
    return H


def gf2_rank(H: np.ndarray) -> int:
    """rank over GF(2) (Gaussian elimination on bit-packed rows): the message length of a code given by a parity-check
    matrix alone is cols - rank -- codes/1920.1280.A lists 5760 checks of rank 1280 for 1920 bits"""
    A = np.packbits(np.asarray(H, np.uint8), axis=1)
    rows, rank = A.shape[0], 0
    for col in range(H.shape[1]):
        byte, bit = col >> 3, 0x80 >> (col & 7)
        hit = np.flatnonzero(A[rank:, byte] & bit)
        if hit.size == 0:
            continue
        piv = rank + int(hit[0])
        if piv != rank:
            A[[rank, piv]] = A[[piv, rank]]
        elim = np.flatnonzero(A[:, byte] & bit)
        elim = elim[elim != rank]
        A[elim] ^= A[rank]
        rank += 1
        if rank == rows:
            break
    return rank


def read_matlab_bits(text: str) -> np.ndarray:
    """Data/BitMatrix/Matlab.hs:20-26: lines of '0'/'1' words."""
    rows = [[int(w) for w in line.split()] for line in text.splitlines() if line.strip()]
    a = np.array(rows, dtype=np.uint8)
    if not np.isin(a, (0, 1)).all():
        raise ValueError("readBit: no parse")
    return a


def read_qc(text: str):
    """QuasiCyclic.hs:52-56: first token = cycle size, rest = Matlab matrix of (big) integers.
    Returns (sz, list-of-lists of Python ints)."""
    lines = [l for l in text.splitlines() if l.strip()]
    sz = int(lines[0].split()[0])
    rest = lines[0].split()[1:]
    rows = []
    if rest:
        rows.append([int(w) for w in rest])
    for l in lines[1:]:
        rows.append([int(w) for w in l.split()])
    w = len(rows[0])
    if any(len(r) != w for r in rows):
        raise ValueError("ragged .q matrix")
    return sz, rows


def qc_expand(sz: int, rows) -> np.ndarray:
    """QuasiCyclic.hs:19-25 toBitMatrix: block (m,n) value v; row i, column j of the block is
    v `testBit` ((j - i) mod sz)."""
    R, C = len(rows), len(rows[0])
    out = np.zeros((R * sz, C * sz), dtype=np.uint8)
    for m in range(R):
        for n in range(C):
            v = rows[m][n]
            if v == 0:
                continue
            for k in range(sz):
                if (v >> k) & 1:
                    i = np.arange(sz)
                    out[m * sz + i, n * sz + (i + k) % sz] = 1
    return out


def qc_offsets(sz: int, rows) -> np.ndarray:
    """Fast/Arraylet.hs:68-79 initMatrixlet (and GPU/CUDA/Arraylet2.hs:299-331): 0 -> -1 (empty),
    one set bit -> its index, anything else is an error in the reference."""
    R, C = len(rows), len(rows[0])
    off = np.full((R, C), -1, dtype=np.int32)
    for m in range(R):
        for n in range(C):
            v = rows[m][n]
            if v == 0:
                continue
            if bin(v).count("1") != 1:
                raise ValueError(f"QuasiCyclic matrix has non-powers of two initial value of {v}")
            off[m, n] = v.bit_length() - 1
    return off


def qc_bits(sz: int, rows) -> np.ndarray:
    """[R][C][sz] bytes: bit b of each block integer (for the QC encoder oracle)."""
    R, C = len(rows), len(rows[0])
    out = np.zeros((R, C, sz), dtype=np.uint8)
    for m in range(R):
        for n in range(C):
            v = rows[m][n]
            for b in range(sz):
                out[m, n, b] = (v >> b) & 1
    return out


def dense_to_csr(H: np.ndarray):
    M, N = H.shape
    row_ptr = np.zeros(M + 1, dtype=np.int32)
    cols = []
    for m in range(M):
        c = np.nonzero(H[m])[0].astype(np.int32)
        cols.append(c)
        row_ptr[m + 1] = row_ptr[m] + len(c)
    col_idx = np.concatenate(cols).astype(np.int32) if cols else np.zeros(0, np.int32)
    return row_ptr, col_idx
