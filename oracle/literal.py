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
