"""ctypes front-end of oracle/ldpc_oracle.c.  TEST INFRASTRUCTURE ONLY -- importable from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never from ecc_ldpc_amd/.  (Four developer scripts under tools/ are
checkers of the same kind and use it as tests do: gen_golden.py (SURVEY.md section 7 places it there), fuzz_qc.py (also run by
tests/test_fuzz_qc_gpu.py), iters_f32_vs_f64.py and layered_f32_vs_f64.py (the measurements the test bars are set from).)"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libldpc_oracle.so")
TANH, MINSUM, TANH_CM = 0, 1, 2
CUDA32 = 3   # the arithmetic of the reference's CUDA plug-in `cuda-arraylet2` (ldpc_oracle.c ORACLE_CUDA32): float state, double products
SUM_ARRAYLET, SUM_SPARSE = 16, 32   # column-sum orders of the other registered decoders (ldpc_oracle.c)
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "ldpc_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        i32p, f64p, u8p = C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_uint8)
        ip = C.POINTER(C.c_int)
        L.oracle_decode.argtypes = [C.c_int, C.c_int, i32p, i32p, C.c_int, C.c_int, f64p, u8p, ip, ip, f64p, f64p, f64p]
        L.oracle_decode_dense.argtypes = [C.c_int, C.c_int, u8p, C.c_int, C.c_int, f64p, u8p, ip, ip, f64p]
        L.oracle_step.argtypes = [C.c_int, C.c_int, i32p, i32p, C.c_int, f64p, f64p, f64p, f64p, f64p, ip]
        L.oracle_decode_batch.argtypes = [C.c_int, C.c_int, i32p, i32p, C.c_int, C.c_int, C.c_int, f64p, u8p, i32p, u8p, C.c_int]
        L.oracle_decode_layered.argtypes = [C.c_int, C.c_int, i32p, i32p, C.c_int, i32p, C.c_int, C.c_int, f64p, u8p, ip, ip, f64p, f64p]
        L.oracle_decode_layered_batch.argtypes = [C.c_int, C.c_int, i32p, i32p, C.c_int, i32p, C.c_int, C.c_int, C.c_int, f64p, u8p, i32p, u8p, C.c_int]
        L.oracle_layered_step.argtypes = [C.c_int, C.c_int, i32p, i32p, C.c_int, f64p, f64p, f64p, f64p, ip, ip]
        L.oracle_encode_dense.argtypes = [C.c_int, C.c_int, u8p, u8p, u8p]
        L.oracle_encode_qc.argtypes = [C.c_int, C.c_int, C.c_int, u8p, u8p, u8p]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def _variant(v):
    if v in (TANH, "tanh"):
        return TANH
    if v in (MINSUM, "min", "minsum", "min-sum"):
        return MINSUM
    if v in (TANH_CM, "cm", "tanh-cm", "arraylet-cm"):
        return TANH_CM
    if v in (CUDA32, "cuda32", "cuda-arraylet2"):
        return CUDA32
    # the other registered decoders: the same check rule, their own column-sum order (ldpc_oracle.c ORACLE_SUM_*)
    if v == "arraylet":
        return TANH | SUM_ARRAYLET
    if v == "arraylet-min":
        return MINSUM | SUM_ARRAYLET
    if v == "sparse":
        return TANH | SUM_SPARSE
    if v == "sparsemin":
        return MINSUM | SUM_SPARSE
    if isinstance(v, int) and (v & 15) in (TANH, MINSUM) and (v & ~15) in (SUM_ARRAYLET, SUM_SPARSE):
        return v
    raise ValueError(v)


class Graph:
    def __init__(self, row_ptr, col_idx, N):
        self.row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32)
        self.col_idx = np.ascontiguousarray(col_idx, dtype=np.int32)
        self.M = len(self.row_ptr) - 1
        self.N = int(N)
        self.E = int(self.row_ptr[-1])

    @classmethod
    def from_dense(cls, H):
        from .formats import dense_to_csr
        rp, ci = dense_to_csr(np.asarray(H))
        return cls(rp, ci, H.shape[1])


def _as_the_variant_stores(a, variant):
    """float64 array of the values the decoder holds: CUDA32 keeps channel LLRs, LLRs and messages as floats (Arraylet2.hs:153 double2Float)"""
    a = np.asarray(a, dtype=np.float64)
    if _variant(variant) == CUDA32:
        a = a.astype(np.float32).astype(np.float64)
    return np.ascontiguousarray(a)


def decode(g: Graph, variant, max_iters, llr, trace=False):
    """-> dict(bits, iters, converged, lam[, trace_lam (iters+1,N), trace_ne (iters,E)])"""
    llr = _as_the_variant_stores(llr, variant)
    assert llr.shape == (g.N,)
    bits = np.zeros(g.N, np.uint8)
    it, cv = C.c_int(0), C.c_int(0)
    lam = np.zeros(g.N, np.float64)
    tl = np.zeros((max_iters + 1, g.N), np.float64) if trace else None
    tn = np.zeros((max(max_iters, 1), g.E), np.float64) if trace else None
    rc = lib().oracle_decode(g.M, g.N, _p(g.row_ptr, C.c_int32), _p(g.col_idx, C.c_int32), _variant(variant),
                             int(max_iters), _p(llr, C.c_double), _p(bits, C.c_uint8), C.byref(it), C.byref(cv),
                             _p(lam, C.c_double), _p(tl, C.c_double), _p(tn, C.c_double))
    if rc != 0:
        raise RuntimeError(f"oracle_decode rc={rc}")
    out = dict(bits=bits, iters=it.value, converged=bool(cv.value), lam=lam)
    if trace:
        out["trace_lam"] = tl[: it.value + 1]
        out["trace_ne"] = tn[: it.value]
    return out


def decode_layered(g: Graph, layer_ptr, variant, max_iters, llr, trace=False):
    """EXTENSION (no reference counterpart): row-layered schedule, rows [layer_ptr[l], layer_ptr[l+1]) form layer l.
    -> dict(bits, iters (sweeps), converged, lam[, trace_lam (iters+1, N): lam after each sweep, row 0 = the input])"""
    llr = np.ascontiguousarray(llr, dtype=np.float64)
    lp = np.ascontiguousarray(layer_ptr, dtype=np.int32)
    bits = np.zeros(g.N, np.uint8)
    it, cv = C.c_int(0), C.c_int(0)
    lam = np.zeros(g.N, np.float64)
    tl = np.zeros((max_iters + 1, g.N), np.float64) if trace else None
    rc = lib().oracle_decode_layered(g.M, g.N, _p(g.row_ptr, C.c_int32), _p(g.col_idx, C.c_int32), len(lp) - 1, _p(lp, C.c_int32),
                                     _variant(variant), int(max_iters), _p(llr, C.c_double), _p(bits, C.c_uint8), C.byref(it), C.byref(cv),
                                     _p(lam, C.c_double), _p(tl, C.c_double))
    if rc != 0:
        raise RuntimeError(f"oracle_decode_layered rc={rc}")
    out = dict(bits=bits, iters=it.value, converged=bool(cv.value), lam=lam)
    if trace:
        out["trace_lam"] = tl[: it.value + 1]
    return out


def decode_layered_batch(g: Graph, layer_ptr, variant, max_iters, llr, nthreads=1):
    llr = np.ascontiguousarray(llr, dtype=np.float64)
    lp = np.ascontiguousarray(layer_ptr, dtype=np.int32)
    F = llr.shape[0]
    bits = np.zeros((F, g.N), np.uint8)
    iters = np.zeros(F, np.int32)
    conv = np.zeros(F, np.uint8)
    rc = lib().oracle_decode_layered_batch(g.M, g.N, _p(g.row_ptr, C.c_int32), _p(g.col_idx, C.c_int32), len(lp) - 1, _p(lp, C.c_int32),
                                           _variant(variant), int(max_iters), F, _p(llr, C.c_double), _p(bits, C.c_uint8),
                                           _p(iters, C.c_int32), _p(conv, C.c_uint8), int(nthreads))
    if rc != 0:
        raise RuntimeError(f"oracle_decode_layered_batch rc={rc}")
    return bits, iters, conv


def layered_step(g: Graph, variant, lam, msg):
    """one teacher-forced sweep of the layered schedule -> (msg', lam', odd, flip)"""
    lam = np.ascontiguousarray(lam, np.float64)
    msg = np.ascontiguousarray(msg, np.float64)
    lam2, msg2 = np.zeros(g.N, np.float64), np.zeros(g.E, np.float64)
    odd, flip = C.c_int(0), C.c_int(0)
    rc = lib().oracle_layered_step(g.M, g.N, _p(g.row_ptr, C.c_int32), _p(g.col_idx, C.c_int32), _variant(variant), _p(lam, C.c_double),
                                   _p(msg, C.c_double), _p(lam2, C.c_double), _p(msg2, C.c_double), C.byref(odd), C.byref(flip))
    if rc != 0:
        raise RuntimeError(f"oracle_layered_step rc={rc}")
    return msg2, lam2, bool(odd.value), bool(flip.value)


def decode_dense(H, variant, max_iters, llr, trace=False):
    H = np.ascontiguousarray(H, dtype=np.uint8)
    M, N = H.shape
    llr = np.ascontiguousarray(llr, dtype=np.float64)
    bits = np.zeros(N, np.uint8)
    it, cv = C.c_int(0), C.c_int(0)
    tl = np.zeros((max_iters + 1, N), np.float64) if trace else None
    rc = lib().oracle_decode_dense(M, N, _p(H, C.c_uint8), _variant(variant), int(max_iters), _p(llr, C.c_double),
                                   _p(bits, C.c_uint8), C.byref(it), C.byref(cv), _p(tl, C.c_double))
    if rc != 0:
        raise RuntimeError(f"oracle_decode_dense rc={rc}")
    out = dict(bits=bits, iters=it.value, converged=bool(cv.value))
    if trace:
        out["trace_lam"] = tl[: it.value + 1]
    return out


def step(g: Graph, variant, orig, lam, ne):
    orig = _as_the_variant_stores(orig, variant)
    lam = _as_the_variant_stores(lam, variant)
    ne = _as_the_variant_stores(ne, variant)
    ne2 = np.zeros(g.E, np.float64)
    lam2 = np.zeros(g.N, np.float64)
    sz = C.c_int(0)
    rc = lib().oracle_step(g.M, g.N, _p(g.row_ptr, C.c_int32), _p(g.col_idx, C.c_int32), _variant(variant),
                           _p(orig, C.c_double), _p(lam, C.c_double), _p(ne, C.c_double), _p(ne2, C.c_double),
                           _p(lam2, C.c_double), C.byref(sz))
    if rc != 0:
        raise RuntimeError(f"oracle_step rc={rc}")
    return ne2, lam2, bool(sz.value)


def decode_batch(g: Graph, variant, max_iters, llr, nthreads=1):
    llr = _as_the_variant_stores(llr, variant)
    F = llr.shape[0]
    assert llr.shape == (F, g.N)
    bits = np.zeros((F, g.N), np.uint8)
    iters = np.zeros(F, np.int32)
    conv = np.zeros(F, np.uint8)
    rc = lib().oracle_decode_batch(g.M, g.N, _p(g.row_ptr, C.c_int32), _p(g.col_idx, C.c_int32), _variant(variant),
                                   int(max_iters), F, _p(llr, C.c_double), _p(bits, C.c_uint8), _p(iters, C.c_int32),
                                   _p(conv, C.c_uint8), int(nthreads))
    if rc != 0:
        raise RuntimeError(f"oracle_decode_batch rc={rc}")
    return bits, iters, conv


def encode_dense(G, msg):
    G = np.ascontiguousarray(G, np.uint8)
    msg = np.ascontiguousarray(msg, np.uint8)
    k, p = G.shape
    par = np.zeros(p, np.uint8)
    rc = lib().oracle_encode_dense(k, p, _p(G, C.c_uint8), _p(msg, C.c_uint8), _p(par, C.c_uint8))
    assert rc == 0
    return par


def encode_qc(sz, gbits, msg):
    gbits = np.ascontiguousarray(gbits, np.uint8)
    R, Cc, s = gbits.shape
    assert s == sz
    msg = np.ascontiguousarray(msg, np.uint8)
    par = np.zeros(Cc * sz, np.uint8)
    rc = lib().oracle_encode_qc(sz, R, Cc, _p(gbits, C.c_uint8), _p(msg, C.c_uint8), _p(par, C.c_uint8))
    assert rc == 0
    return par
