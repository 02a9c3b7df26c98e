"""Shared test plumbing: loads the shipped matrices through the ORACLE-side parsers, builds
frames, and defines the LLR tolerance used by the floating-point parity tests."""
from __future__ import annotations

import functools
import os

import numpy as np

from oracle import channel, formats, oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODES = os.path.join(ROOT, "codes")

# name -> (k, n_tx) used for the channel (rate 4/5 puncturing for the AR4JA codes, Utils.hs:46-51)
CODE_PARAMS = {
    "moon.7.13": (7, 20),
    "jpl.1024.4.5": (1024, 1280),
    "jpl.4096.4.5": (4096, 5120),
    "1920.1280.3.303": (640, 1920),
}


class LoadedCode:
    def __init__(self, name):
        self.name = name
        self.sz = 0
        self.offsets = None
        self.G = None
        self.gq = None
        d = os.path.join(CODES, name)
        if name.startswith("jpl."):
            self.sz, rows = formats.read_qc(open(os.path.join(d, "H.q")).read())
            self.offsets = formats.qc_offsets(self.sz, rows)
            self.H = formats.qc_expand(self.sz, rows)
            gs, grows = formats.read_qc(open(os.path.join(d, "G.q")).read())
            self.gq = (gs, formats.qc_bits(gs, grows))
        elif name == "moon.7.13":
            self.H = formats.read_alist_reference(open(os.path.join(d, "H.alist")).read())
            self.G = formats.read_alist_reference(open(os.path.join(d, "G.alist")).read())
        elif name == "1920.1280.3.303":
            self.H = formats.read_alist_mackay(open(d).read())
        else:
            raise KeyError(name)
        self.M, self.N = self.H.shape
        self.k, self.n_tx = CODE_PARAMS[name]
        self.graph = oracle.Graph.from_dense(self.H)
        self.E = self.graph.E

    def encode(self, msg):
        if self.gq is not None:
            return np.concatenate([msg, oracle.encode_qc(self.gq[0], self.gq[1], msg)])
        if self.G is not None:
            return np.concatenate([msg, oracle.encode_dense(self.G, msg)])
        return np.zeros(self.N, np.uint8)  # no G shipped: all-zero codeword (valid for a linear code)

    def frames(self, F, ebn0_db, seed):
        rng = np.random.default_rng(seed)
        if self.gq is None and self.G is None:
            cws = np.zeros((F, self.N), np.uint8)
        else:
            cws = np.stack([self.encode(rng.integers(0, 2, self.k).astype(np.uint8)) for _ in range(F)])
        llr = channel.frames(cws, ebn0_db, self.k, self.n_tx, self.N, seed + 1)
        return cws, llr

    def hip_code(self, E, prefer_qc=True):
        if self.offsets is not None and prefer_qc:
            return E.Code.from_qc(self.sz, self.offsets)
        return E.Code.from_csr(self.graph.row_ptr, self.graph.col_idx, self.N)


@functools.lru_cache(maxsize=None)
def load(name) -> LoadedCode:
    return LoadedCode(name)


def lam_tolerance(g, ne_ref, lam_ref, rel=1e-5):
    """Per-entry tolerance for an fp32 LLR compared with the double oracle:
        rel * max(1, |lam_ref|)                      (north_star: 1e-5)
      + the oracle's OWN rounding uncertainty: a double tanh product within 2^-53 * d of +-1 turns
        that rounding into d * 2^-53 / (1 - |p|) = (d/2) * 2^-53 * exp(|ne|) absolute error of the
        returned message (atanh is ill-conditioned there).  It is
        negligible (< 1e-9) for |ne| < 20 and reaches O(1) only next to the 37.43 clamp."""
    d = np.diff(g.row_ptr).max()
    edge_unc = 0.5 * d * 2.0 ** -53 * np.exp(np.minimum(np.abs(ne_ref), 40.0)) * 4.0
    col_unc = np.zeros(g.N)
    np.add.at(col_unc, g.col_idx, edge_unc)
    return rel * np.maximum(1.0, np.abs(lam_ref)) + col_unc, rel * np.maximum(1.0, np.abs(ne_ref)) + edge_unc
