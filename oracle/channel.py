"""Synthetic-frame generator used by tests and the golden-vector tool (TEST INFRASTRUCTURE).
The reference's channel lives in ecc-manifold (absent) -- this is the
build's own stated model: BPSK bit b -> 2b-1 (LLR > 0 <=> bit 1, matching
hard x = x > 0), noise N(0, sigma^2), sigma^2 = 1 / (2 R 10^(EbN0/10)), R = k / n_tx,
LLR = 2 y / sigma^2, punctured tail = 0.0.  LLRs are rounded to float32 so that the fp32
device path and the double oracle see identical inputs."""
from __future__ import annotations

import numpy as np


def sigma2(ebn0_db: float, k: int, n_tx: int) -> float:
    return 1.0 / (2.0 * (k / n_tx) * 10.0 ** (ebn0_db / 10.0))


def frames(codewords: np.ndarray, ebn0_db: float, k: int, n_tx: int, N: int, seed: int) -> np.ndarray:
    """codewords: [F][>=n_tx] 0/1 -> float32-exact LLRs [F][N] as float64."""
    rng = np.random.default_rng(seed)
    F = codewords.shape[0]
    s2 = sigma2(ebn0_db, k, n_tx)
    x = 2.0 * codewords[:, :n_tx].astype(np.float64) - 1.0
    y = x + rng.normal(0.0, np.sqrt(s2), size=(F, n_tx))
    llr = np.zeros((F, N), dtype=np.float64)
    llr[:, :n_tx] = (2.0 * y / s2).astype(np.float32).astype(np.float64)
    return llr
