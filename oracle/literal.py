"""Literal pure-Python/NumPy transliteration of the reference decoder loop, kept deliberately
independent of ldpc_oracle.c (different language, dense indexing, Python floats == C doubles,
math.tanh/math.log == the libm calls GHC makes).  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED
(see ldpc_oracle.c header).  Used by tools/gen_golden.py and tests/test_oracle.py to pin the
C oracle bit-for-bit on small cases.

Follows /root/reference/src/ECC/Code/LDPC/Reference/Orig.hs:58-98 (tanh) and
Reference/Min.hs:54-104 (min-sum) line by line; Utils.hs:113-117 for atanh'.
"""
from __future__ import annotations

import math

import numpy as np

ATANH_CLAMP = 18.714973875118524  # Utils.hs:115


def signum(x: float) -> float:
    return 1.0 if x > 0 else (-1.0 if x < 0 else x)


def atanh_prime(x: float) -> float:
    """Utils.hs:113-117 over base-4.9's  atanh x = 0.5 * log ((1.0+x) / (1.0-x))."""
    den = 1.0 - x
    if den == 0.0:
        y = math.inf if (1.0 + x) > 0 else -math.inf  # (1+x)/0 = inf ; log inf = inf
    else:
        q = (1.0 + x) / den
        y = 0.5 * math.log(q) if q > 0 else (-math.inf if q == 0 else math.nan)
    if math.isinf(y):
        return signum(x) * ATANH_CLAMP
    return y


def min_prime(x: float, y: float) -> float:  # Min.hs:82
    return signum(x) * signum(y) * min(abs(x), abs(y))


def hard(x: float) -> bool:  # GPU/Reference.hs:59-60
    return x > 0


def ldpc(H: np.ndarray, variant: str, max_iterations: int, orig_lam, trace=None):
    """Returns (bits, iterations_run, converged).  `trace`, if a list, receives lam at the top of
    every loop turn (Orig.hs:67-71)."""
    M, N = H.shape
    rows = [list(np.nonzero(H[m])[0]) for m in range(M)]          # ascending j, Orig.hs:88
    orig = [float(v) for v in orig_lam]
    lam = list(orig)
    ne = [[0.0] * N for _ in range(M)]                              # Orig.hs:64-65
    n = 0
    while True:
        if trace is not None:
            trace.append(list(lam))
        ans = [0] * M                                               # Orig.hs:77-78
        for m in range(M):
            p = 0
            for j in rows[m]:
                p ^= 1 if hard(lam[j]) else 0
            ans[m] = p
        if not any(ans):
            return np.array([hard(v) for v in lam], dtype=np.uint8), n, True
        if n >= max_iterations:
            return np.array([hard(v) for v in orig], dtype=np.uint8), n, False
        ne2 = [[0.0] * N for _ in range(M)]
        for m in range(M):
            for c in rows[m]:
                if variant == "tanh":
                    prod = 1.0
                    for j in rows[m]:
                        if j != c:
                            prod = prod * math.tanh(-((lam[j] - ne[m][j]) / 2.0))
                    ne2[m][c] = -2.0 * atanh_prime(prod)
                else:
                    xs = [-(lam[j] - ne[m][j]) for j in rows[m] if j != c]
                    if not xs:
                        raise ValueError("foldr1: empty list")
                    acc = xs[-1]
                    for x in reversed(xs[:-1]):                      # foldr1 min'
                        acc = min_prime(x, acc)
                    ne2[m][c] = (-3.0 / 4.0) * acc
        lam2 = [0.0] * N
        for j in range(N):                                          # Orig.hs:96 foldr (+) orig
            acc = orig[j]
            for m in range(M - 1, -1, -1):
                acc = ne2[m][j] + acc
            lam2[j] = acc
        ne, lam = ne2, lam2
        n += 1


# ---------------------------------------------------------------------------------------------------------------------
# `arraylet-cm` (src/ECC/Code/LDPC/Fast/CachedMult.hs): the tanh rule with the row product cached as a StableDiv
def _lit(x):                       # CachedMult.hs:41-44
    return (1.0, x) if x >= 1 else (x, 1.0)


def _abs_min_max(x, y):            # :31-34
    return (x, y) if abs(x) < abs(y) else (y, x)


def _smult(p, q):                  # :46-50
    (a, b), (c, d) = p, q
    mn, mx = _abs_min_max(a, c)
    return (mn, b * mx * d)


def _sdiv(p, c):                   # :52-55
    a, b = p
    return b if a == c else a * (b / c)


def ldpc_cm(sz: int, offsets: np.ndarray, max_iterations: int, orig_lam, trace=None):
    """CachedMult.hs:233-264 over the Matrixlet of a quasi-cyclic H (offsets[br][bc] = rotation or -1), written the
    way the Haskell folds run: per block row a foldr1 over its non-empty blocks in ascending block column of
    element-wise smult (:184-188), per block column a foldr1 (+) over ascending block rows (:190-194)."""
    R, Cb = offsets.shape
    N = Cb * sz
    orig = [float(v) for v in orig_lam]
    lam = list(orig)
    ne = {(br, bc): [0.0] * sz for br in range(R) for bc in range(Cb) if offsets[br, bc] >= 0}   # indexed by row r'
    col = lambda bc, br, r: bc * sz + (r + int(offsets[br, bc])) % sz                             # :84 arrayArraylet
    n = 0
    while True:
        if trace is not None:
            trace.append(list(lam))
        ok = True
        for br in range(R):                       # ans (:244-245): foldr1 (zipWith (/=)) over the row's blocks
            blocks = [bc for bc in range(Cb) if offsets[br, bc] >= 0]
            for r in range(sz):
                p = False
                for bc in blocks:
                    p ^= hard(lam[col(bc, br, r)])
                ok = ok and not p
        if ok:
            return np.array([hard(v) for v in lam], np.uint8), n, True
        if n >= max_iterations:
            return np.array([hard(v) for v in orig], np.uint8), n, False
        th = {k: [math.tanh(-((lam[col(k[1], k[0], r)] - v[r]) / 2)) for r in range(sz)] for k, v in ne.items()}   # :247-251
        ne2 = {}
        for br in range(R):
            blocks = [bc for bc in range(Cb) if offsets[br, bc] >= 0]
            for r in range(sz):
                acc = _lit(th[(br, blocks[-1])][r])
                for bc in reversed(blocks[:-1]):                       # foldr1: l1 `smult` (l2 `smult` (... ld))
                    acc = _smult(_lit(th[(br, bc)][r]), acc)
                for bc in blocks:                                      # :256-259
                    ne2.setdefault((br, bc), [0.0] * sz)[r] = -2 * atanh_prime(_sdiv(acc, th[(br, bc)][r]))
        lam2 = list(orig)
        for bc in range(Cb):
            rows = [br for br in range(R) if offsets[br, bc] >= 0]
            for c in range(sz):                                        # foldRowsArraylet: column c <- row (c - off) mod sz
                vals = [ne2[(br, bc)][(c - int(offsets[br, bc])) % sz] for br in rows]
                acc = vals[-1]
                for v in reversed(vals[:-1]):
                    acc = v + acc
                lam2[bc * sz + c] = orig[bc * sz + c] + acc            # :261-262 zipWith (+) orig_lam
        ne, lam, n = ne2, lam2, n + 1


# ---------------------------------------------------------------------------------------------------------------------
# EXTENSION (no reference counterpart): the row-layered schedule specified at oracle_decode_layered in ldpc_oracle.c
def ldpc_layered(H: np.ndarray, layer_ptr, variant: str, max_iterations: int, orig_lam, trace=None):
    M, N = H.shape
    rows = [list(np.nonzero(H[m])[0]) for m in range(M)]
    orig = [float(v) for v in orig_lam]
    lam = list(orig)
    msg = [[0.0] * len(rows[m]) for m in range(M)]
    if trace is not None:
        trace.append(list(lam))
    if all(sum(hard(lam[j]) for j in rows[m]) % 2 == 0 for m in range(M)):
        return np.array([hard(v) for v in lam], np.uint8), 0, True
    n = 0
    while True:
        if n >= max_iterations:
            return np.array([hard(v) for v in orig], np.uint8), n, False
        odd = flip = False
        for l in range(len(layer_ptr) - 1):
            for m in range(layer_ptr[l], layer_ptr[l + 1]):
                cs = rows[m]
                t = [lam[c] - msg[m][k] for k, c in enumerate(cs)]
                odd = odd or (sum(hard(lam[c]) for c in cs) % 2 == 1)
                if variant == "tanh":
                    x = [math.tanh(-(v / 2)) for v in t]
                    new = []
                    for k in range(len(cs)):
                        prod = 1.0
                        for j in range(len(cs)):
                            if j != k:
                                prod = prod * x[j]
                        new.append(-2 * atanh_prime(prod))
                else:
                    x = [-v for v in t]
                    new = []
                    for k in range(len(cs)):
                        others = [x[j] for j in range(len(cs)) if j != k]
                        acc = others[-1]
                        for v in reversed(others[:-1]):
                            acc = min_prime(v, acc)
                        new.append((-3 / 4) * acc)
                for k, c in enumerate(cs):
                    nw = t[k] + new[k]
                    flip = flip or (hard(nw) != hard(lam[c]))
                    lam[c] = nw
                    msg[m][k] = new[k]
        n += 1
        if trace is not None:
            trace.append(list(lam))
        if not odd and not flip:
            return np.array([hard(v) for v in lam], np.uint8), n, True


# ---------------------------------------------------------------------------------------------------------------------
# The other registered decoders of the reference: the same check rule as Orig.hs / Min.hs, their OWN column-sum order.
def ldpc_arraylet(sz: int, offsets: np.ndarray, variant: str, max_iterations: int, orig_lam, trace=None):
    """Fast/Arraylet.hs:164-186 (`arraylet`, variant "tanh") and Fast/ArrayletMin.hs:167-192 (`arraylet-min`, variant "min")
    over the Matrixlet of a quasi-cyclic H (offsets[br][bc] = rotation or -1), written the way the Haskell folds run:
    ne_tanh per row = foldr1 (zipWith (++)) over the row's blocks in ascending block column (Arraylet.hs:99-103), i.e. the
    row's (column, value) pairs in ascending column; the rule over `[v | (j, v) <- ne_tanh ! m, j /= n]`; and
    lam' = zipWith (+) orig_lam (foldRowsMatrixlet (+) ne') -- per column a foldr1 (+) over ascending block rows
    (Arraylet.hs:105-109), THEN orig + that."""
    R, Cb = offsets.shape
    orig = [float(v) for v in orig_lam]
    lam = list(orig)
    ne = {(br, bc): [0.0] * sz for br in range(R) for bc in range(Cb) if offsets[br, bc] >= 0}   # indexed by row r'
    col = lambda bc, br, r: bc * sz + (r + int(offsets[br, bc])) % sz                             # Arraylet.hs:45 arrayArraylet
    n = 0
    while True:
        if trace is not None:
            trace.append(list(lam))
        ok = True
        for br in range(R):                       # ans: foldr1 (zipWith (/=)) over the row's blocks
            blocks = [bc for bc in range(Cb) if offsets[br, bc] >= 0]
            for r in range(sz):
                p = False
                for bc in blocks:
                    p ^= hard(lam[col(bc, br, r)])
                ok = ok and not p
        if ok:
            return np.array([hard(v) for v in lam], np.uint8), n, True
        if n >= max_iterations:
            return np.array([hard(v) for v in orig], np.uint8), n, False
        ne2 = {}
        for br in range(R):
            blocks = [bc for bc in range(Cb) if offsets[br, bc] >= 0]
            for r in range(sz):
                if variant == "tanh":   # Arraylet.hs:179-183
                    row = [(col(bc, br, r), math.tanh(-((lam[col(bc, br, r)] - ne[(br, bc)][r]) / 2))) for bc in blocks]
                    for bc in blocks:
                        prod = 1.0
                        for j, v in row:                         # product: a left fold from 1
                            if j != col(bc, br, r):
                                prod = prod * v
                        ne2.setdefault((br, bc), [0.0] * sz)[r] = -2 * atanh_prime(prod)
                else:                   # ArrayletMin.hs:182-189
                    row = [(col(bc, br, r), -(lam[col(bc, br, r)] - ne[(br, bc)][r])) for bc in blocks]
                    for bc in blocks:
                        vals = [v for j, v in row if j != col(bc, br, r)]
                        acc = vals[-1]
                        for v in reversed(vals[:-1]):            # foldr1 min'
                            acc = min_prime(v, acc)
                        ne2.setdefault((br, bc), [0.0] * sz)[r] = (-3 / 4) * acc
        lam2 = list(orig)
        for bc in range(Cb):
            rows = [br for br in range(R) if offsets[br, bc] >= 0]
            for c in range(sz):                                  # foldRowsArraylet: column c <- row (c - off) mod sz
                vals = [ne2[(br, bc)][(c - int(offsets[br, bc])) % sz] for br in rows]
                acc = vals[-1]
                for v in reversed(vals[:-1]):                    # foldr1 (+)
                    acc = v + acc
                lam2[bc * sz + c] = orig[bc * sz + c] + acc      # zipWith (+) orig_lam
        ne, lam, n = ne2, lam2, n + 1


def ldpc_sparse(H: np.ndarray, variant: str, max_iterations: int, orig_lam, trace=None):
    """Reference/Sparse.hs:46-115 (`sparse`, variant "tanh") and Reference/SparseMin.hs:49-119 (`sparsemin`, variant "min"):
    messages in a column-major assoc structure; the rule over `j <- ones, j /= n` (the row's columns, ascending:
    Sparse.hs:71-75 rowOnes); lam' = orig_lam ! j + colSum j ne', colSum = sum . map snd over the column's (row, value) list
    (Data/Sparse/Matrix.hs:35-36) -- a left fold from 0 over ascending rows (the list is built row by row, :48-52 with the
    index list of Sparse.hs colMajorIndexList)."""
    M, N = H.shape
    orig = [float(v) for v in orig_lam]
    lam = list(orig)
    ones = [[j for j in range(N) if H[m, j]] for m in range(M)]
    col_rows = [[m for m in range(M) if H[m, j]] for j in range(N)]
    ne = {}
    at = lambda m, j: ne.get((m, j), 0.0)                                   # (!!!): absent -> 0
    n = 0
    while True:
        if trace is not None:
            trace.append(list(lam))
        c_hat = [hard(v) for v in lam]
        if all(sum(c_hat[j] for j in ones[m]) % 2 == 0 for m in range(M)):  # allMultFalse
            return np.array(c_hat, np.uint8), n, True
        if n >= max_iterations:
            return np.array([hard(v) for v in orig], np.uint8), n, False
        ne2 = {}
        for m in range(M):
            for nn in ones[m]:
                if variant == "tanh":
                    prod = 1.0
                    for j in ones[m]:
                        if j != nn:
                            prod = prod * math.tanh(-((lam[j] - at(m, j)) / 2))
                    ne2[(m, nn)] = -2 * atanh_prime(prod)
                else:
                    vals = [-(lam[j] - at(m, j)) for j in ones[m] if j != nn]
                    acc = vals[-1]
                    for v in reversed(vals[:-1]):
                        acc = min_prime(v, acc)
                    ne2[(m, nn)] = (-3 / 4) * acc
        lam2 = []
        for j in range(N):
            s = 0.0
            for m in col_rows[j]:                                           # sum = foldl (+) 0
                s = s + ne2[(m, j)]
            lam2.append(orig[j] + s)
        ne, lam, n = ne2, lam2, n + 1


# ---------------------------------------------------------------------------------------------------------------------
# The reference's LIVE GPU decoder `cuda-arraylet2`, kernel by kernel, thread by thread (GPU/CUDA/Arraylet2.hs:153-279 driving
# cudabits/arraylet2.cu:43-83 and cudabits/common.h:82-97,151-247; float_ty = float).  A literal walk over its arrays -- mLet
# [rowCount][colCount] floats, lam [colCount * sz] floats -- with numpy float32 / float64 scalars standing for the C types.
def _libm_atanhf():
    import ctypes
    import ctypes.util
    fn = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6").atanhf
    fn.restype, fn.argtypes = ctypes.c_float, [ctypes.c_float]
    return fn


def ldpc_cuda_arraylet2(sz: int, offsets: np.ndarray, max_iterations: int, orig_lam, trace=None):
    f32, f64 = np.float32, np.float64
    atanhf = _libm_atanhf()                              # the C library's float atanh, as ldpc_oracle.c calls it (CUDA's differs in last ulps)
    R, colCount = offsets.shape
    rowCount = R * sz
    orig = [f32(v) for v in orig_lam]                   # Arraylet2.hs:153 double2Float
    lam = list(orig)                                    # :160 poke orig_lam -> lam_dev
    mLet = [[f32(0)] * colCount for _ in range(rowCount)]   # :166 memset 0

    def lamIndex(i, j):                                 # common.h:90-98
        off = int(offsets[j // sz, i])
        return i * sz + (off + j) % sz if off > -1 else -1

    def atanh_(x):                                      # common.h:82-88 (x: float)
        x = f32(x)
        if x == 1 or x == -1:
            return f32(f64(-1 if x < 0 else 1) * f64(18.714973875118524))
        return f32(atanhf(float(x)))                    # float atanh

    iters = 0
    while True:
        if trace is not None:
            trace.append([float(v) for v in lam])
        if iters >= max_iterations:                     # Arraylet2.hs:165
            return np.array([v > 0 for v in orig], np.uint8), iters, False
        done = 0                                        # parityRowResults, common.h:212-233
        for j in range(rowCount):
            count = sum(1 for i in range(colCount) if lamIndex(i, j) > -1 and lam[lamIndex(i, j)] > 0)
            if count % 2 == 1:
                done = 1
        if not done:                                    # Arraylet2.hs:198 `when parity`: otherwise the result is lam
            return np.array([v > 0 for v in lam], np.uint8), iters, True
        newMLet = [[f32(0)] * colCount for _ in range(rowCount)]
        for j in range(rowCount):                       # selfProduct, arraylet2.cu:43-83: one thread per (i, j)
            smem = [f32(1)] * colCount
            for i in range(colCount):
                v = f64(mLet[j][i])                     # :51 double v
                ix = lamIndex(i, j)
                if ix > -1:
                    smem[i] = f32(math.tanh(-((float(lam[ix]) - float(v)) / 2.0)))   # :56: float - double -> double tanh (libm), stored as float
            for i in range(colCount):
                if lamIndex(i, j) > -1:
                    r = f64(1)                          # :49
                    for k in range(0, i):
                        r = r * f64(smem[k])            # :73-75
                    for k in range(i + 1, colCount):
                        r = r * f64(smem[k])            # :77-79
                    newMLet[j][i] = f32(-2) * atanh_(r)  # :81
        lam = list(orig)                                # Arraylet2.hs:239 copyArray orig_lam_dev lam_dev
        mLet = newMLet                                  # :241 swapRefs
        for i in range(colCount * sz):                  # updateLam, common.h:151-172
            blockI = i // sz
            for j in range(rowCount // sz):
                if int(offsets[j, blockI]) < 0:         # (:161-167: the guard the C intends -- with off = sz - (-1) its `off > -1` is always true
                    continue                            #  and an absent block adds the 0 that selfProduct left there: same value)
                off = sz - int(offsets[j, blockI])
                localR = (i + off) % sz
                r = j * sz + localR
                lam[i] = f32(lam[i] + mLet[r][blockI])
        iters += 1
