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
    "1920.1280.A": (640, 1920),          # 5760 redundant checks of rank 1280
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
        elif name in ("1920.1280.3.303", "1920.1280.A"):
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


def iters_agree(its, oi):
    """f32 kernel vs Double oracle: hard bits and flags are compared exactly elsewhere; the TURN a frame stops at may move by one
    when a float LLR and the Double LLR straddle zero.  Measured (tools/iters_f32_vs_f64.py, profiles/r03_iters_f32_vs_f64.txt):
    14 398 of 14 400 frames identical over three codes, both rules, the waterfall; the two others off by +1 / -1.  Bar: at most
    max(1, 0.5 %) of the frames differ, none by more than one turn."""
    d = np.asarray(its).astype(int) - np.asarray(oi).astype(int)
    return int((d != 0).sum()) <= max(1, int(0.005 * len(d))) and (np.abs(d).max() if len(d) else 0) <= 1


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


# ---------------------------------------------------------------------------------------------------------------------
# Synthetic quasi-cyclic codes for the run-time specialised kernels (any single-circulant .q is a valid input of the
# reference's QC decoders, Fast/Arraylet.hs:68-79).  Deterministic; frames are the all-zero codeword + noise.
class SyntheticQC:
    def __init__(self, name, sz, offsets, rate=None):
        self.name, self.sz = name, int(sz)
        self.offsets = np.asarray(offsets, np.int32)
        R, Cb = self.offsets.shape
        self.M, self.N = R * sz, Cb * sz
        H = np.zeros((self.M, self.N), np.uint8)
        r = np.arange(sz)
        for br in range(R):
            for bc in range(Cb):
                if self.offsets[br, bc] >= 0:
                    H[br * sz + r, bc * sz + (r + self.offsets[br, bc]) % sz] = 1     # QuasiCyclic.hs:19-25
        self.H = H
        self.graph = oracle.Graph.from_dense(H)
        self.E = self.graph.E
        self.k, self.n_tx = rate if rate else (self.N - self.M, self.N)
        self.layer_ptr = np.arange(0, self.M + 1, sz, dtype=np.int32)                   # one block row per layer

    def frames(self, F, ebn0_db, seed):
        cws = np.zeros((F, self.N), np.uint8)
        return cws, channel.frames(cws, ebn0_db, self.k, self.n_tx, self.N, seed)

    def hip_code(self, E):
        return E.Code.from_qc(self.sz, self.offsets)


def _random_offsets(mask, sz, seed):
    rng = np.random.default_rng(seed)
    return np.where(mask, rng.integers(0, sz, mask.shape), -1).astype(np.int32)


@functools.lru_cache(maxsize=None)
def synthetic(name) -> SyntheticQC:
    if name == "jpl4096-permuted":          # the headline code's block structure with other rotations
        base = load("jpl.4096.4.5").offsets
        return SyntheticQC(name, 128, _random_offsets(base >= 0, 128, 11), rate=(4096, 5120))
    if name == "regular36-sz128":           # (3,6)-regular protograph, 6 x 12 blocks
        rng = np.random.default_rng(12)
        mask = np.zeros((6, 12), bool)
        for bc in range(12):                # column weight 3, row weight 6
            for j in range(3):
                mask[(bc + 2 * j + (bc // 6)) % 6, bc] = True
        assert (mask.sum(0) == 3).all() and (mask.sum(1) == 6).all()
        return SyntheticQC(name, 128, _random_offsets(mask, 128, 13))
    if name == "ira-12x24-sz64":            # rate-1/2 irregular repeat-accumulate shape (dual-diagonal parity part)
        rng = np.random.default_rng(14)
        mask = np.zeros((12, 24), bool)
        for br in range(12):
            mask[br, 12 + br] = True
            if br:
                mask[br, 12 + br - 1] = True
        mask[0, 23] = True
        for bc in range(12):
            w = 6 if bc < 4 else 3
            for br in rng.choice(12, w, replace=False):
                mask[br, bc] = True
        off = _random_offsets(mask, 64, 15)
        for br in range(12):                # accumulator: rotation 0 on the two diagonals
            off[br, 12 + br] = 0
            if br:
                off[br, 12 + br - 1] = 0
        return SyntheticQC(name, 64, off)
    if name == "small-2x4-sz32":            # two frames per wave (sz < 64), one wave group
        return SyntheticQC(name, 32, np.array([[1, 7, 30, -1], [5, -1, 12, 3]], np.int32))
    if name == "irregular-20x30-sz64":      # 20 x 30 blocks, block-row weights 3..10: needs three or more wave groups
        rng = np.random.default_rng(16)
        mask = np.zeros((20, 30), bool)
        for br in range(20):
            for bc in rng.choice(30, 3 + (br * 7) % 8, replace=False):
                mask[br, bc] = True
        for bc in range(30):
            if not mask[:, bc].any():
                mask[rng.integers(0, 20), bc] = True
        return SyntheticQC(name, 64, _random_offsets(mask, 64, 17))
    if name == "wide-4x40-sz256":           # circulant size 256: four waves per wave group
        rng = np.random.default_rng(18)
        mask = np.zeros((4, 40), bool)
        for bc in range(40):
            for br in rng.choice(4, 2 + (bc % 2), replace=False):
                mask[br, bc] = True
        return SyntheticQC(name, 256, _random_offsets(mask, 256, 19))
    if name in ("wimax-12x24-sz96", "wifi-12x24-sz27", "dvbs2short-20x45-sz360"):
        # circulant sizes that are NOT powers of two (WiMAX 802.16e: 24..96, WiFi 802.11n: 27/54/81, DVB-S2: 360): IRA shapes,
        # synthetic rotations.  dvbs2short: n = 16 200, k = 9 000, the short-frame size of DVB-S2 (its LLRs fit in LDS: 65 KB)
        sz = {"wimax-12x24-sz96": 96, "wifi-12x24-sz27": 27, "dvbs2short-20x45-sz360": 360}[name]
        R, K = (20, 25) if sz == 360 else (12, 12)
        rng = np.random.default_rng(20 + sz)
        mask = np.zeros((R, K + R), bool)
        for br in range(R):
            mask[br, K + br] = True
            if br:
                mask[br, K + br - 1] = True
        mask[0, K + R - 1] = True
        for bc in range(K):
            for br in rng.choice(R, 3 + (3 if bc % 4 == 0 else 0), replace=False):
                mask[br, bc] = True
        off = _random_offsets(mask, sz, 21 + sz)
        for br in range(R):
            off[br, K + br] = 0
            if br:
                off[br, K + br - 1] = 0
        return SyntheticQC(name, sz, off)
    if name in ("latin-24x16-sz64", "latin-18x9-sz40"):
        # block rows in GROUPS whose members share no block column (each group partitions the block columns): what csrc/layered_lds.hip
        # runs together.  sz64: six groups of four (four waves of one block row each); sz40: six groups of three, idle lanes in every wave
        Rg, Cs, sz = (4, 4, 64) if name.startswith("latin-24") else (3, 3, 40)
        nbc, NG = Rg * Cs, 6                 # six groups (the pipelined loop wants more groups than it keeps records in flight)
        rng = np.random.default_rng(22 + sz)
        mask = np.zeros((NG * Rg, nbc), bool)
        for gi in range(NG):                 # every group deals the block columns out to its block rows: Cs each, none shared
            perm = rng.permutation(nbc)
            for c in range(nbc):
                mask[gi * Rg + perm[c] // Cs, c] = True
        assert all(not (mask[gi * Rg + a] & mask[gi * Rg + b]).any() for gi in range(NG) for a in range(Rg) for b in range(a))
        return SyntheticQC(name, sz, _random_offsets(mask, sz, 23 + sz), rate=(nbc * sz // 4, nbc * sz))   # (M >= N here: a nominal rate for the channel)
    if name == "heavycol-24x8-sz64":
        # a QC form of what codes/1920.1280.A is: every check of a (3,6)-regular code (4 x 8 blocks) written SIX times -- block-row
        # weight 6, block-column weight 18, rank that of the 4 x 8 matrix.  The six copies of a check send the same message, so
        # flooding min-sum multiplies the LLRs of a frame that does not converge by several units per turn: past FLT_MAX well
        # before turn 50 -- the overflow case of the QC kernels (tests/test_overflow_gpu.py)
        base = np.ones((4, 8), bool)
        for b in range(4):
            base[b, 2 * b] = base[b, 2 * b + 1] = False
        off4 = _random_offsets(base, 64, 97)
        return SyntheticQC(name, 64, np.concatenate([off4] * 6), rate=(256, 512))
    raise KeyError(name)


# codes whose run-time specialised kernels __graft_entry__.build() also prepares (name -> [(variant, dtype, schedule)])
EXTRA_JIT = {"heavycol-24x8-sz64": [("min", "f32", "flooding"), ("min", "f32", "layered")]}
SYNTHETIC_NAMES = ["jpl4096-permuted", "regular36-sz128", "ira-12x24-sz64", "small-2x4-sz32", "irregular-20x30-sz64", "wide-4x40-sz256",
                   "wimax-12x24-sz96", "wifi-12x24-sz27", "dvbs2short-20x45-sz360"]
