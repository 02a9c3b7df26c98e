"""oracle/emulate_f16.py -- TEST INFRASTRUCTURE ONLY (imported by tests/, never by the product).

NumPy emulation of the LDPC_F16 decoders ("fp16 storage in HBM, fp32 arithmetic", include/ldpc_hip.h).
The reference has no fp16 path (its CUDA plug-ins compute in float32, cudabits/arraylet2.cu:43-83), so this
is the build's own extension with its own oracle -- SURVEY.md section 8f item 4 -- PARITY UNPINNED against
the reference by construction.  What it pins is that the HIP kernels do exactly what the header says:

  flood path : the loop of Reference/Min.hs:54-104 (syndrome -> return lam | out of turns -> return orig |
               check-node rule (-3/4) * foldr1 min', column sum foldr (+) orig in descending row order),
               with EVERY stored quantity -- channel LLRs, messages, lam -- saturated to +-65504 and rounded
               to fp16 (nearest even) when written, and all arithmetic on the float32 values read back.
               float32 add/sub/mul/min/compare are exactly rounded IEEE operations on both sides (the
               library is built with -ffp-contract=off), so the min-sum trajectory is reproduced BIT FOR BIT.
  fused paths: the f32 decoder fed r16(llr); only the channel LLRs ever live in HBM.  (Tested as
               fused-F16(llr) == fused-F32(r16(llr)); no separate emulation needed.)
"""
import numpy as np

F16_MAX = np.float32(65504.0)


def r16(x):
    """value of float32 x after being stored as fp16: saturate, round to nearest even, read back"""
    x = np.asarray(x, np.float32)
    return np.clip(x, -F16_MAX, F16_MAX).astype(np.float16).astype(np.float32)


def _rows(g):
    return [(m, g.col_idx[g.row_ptr[m]:g.row_ptr[m + 1]], int(g.row_ptr[m])) for m in range(g.M)]


def cn_minsum_f32(t):
    """t [F, D] float32 = lam_j - ne[m,j]  ->  ne'[m,:]  (ldpc_math.h cn_minsum; Min.hs:78-86)"""
    a = np.abs(t)
    pos = t > 0                                   # x_j = -t_j is negative <=> t_j > 0
    par = np.logical_xor.reduce(pos, axis=1, keepdims=True)
    i1 = np.argmin(a, axis=1)                     # first minimum, as the kernel's strict '<' scan keeps it
    m1 = np.take_along_axis(a, i1[:, None], axis=1)
    a2 = a.copy()
    np.put_along_axis(a2, i1[:, None], np.float32(np.inf), axis=1)
    m2 = a2.min(axis=1, keepdims=True)
    k = np.arange(t.shape[1])[None, :]
    mag = np.where(k == i1[:, None], m2, m1)
    neg = np.logical_xor(par, pos)
    acc = np.where(neg, -mag, mag).astype(np.float32)
    return (np.float32(-0.75) * acc).astype(np.float32)


def step_minsum_f16_flood(g, orig, lam, ne):
    """one update of the flood F16 decoder; all arrays float32 holding fp16-representable values.
    orig, lam [F, N]; ne [F, E] in CSR edge order  ->  ne', lam', syndrome_zero [F]"""
    F = lam.shape[0]
    hard = lam > 0
    ne2 = np.empty_like(ne)
    syn_ok = np.ones(F, bool)
    for m, cols, e0 in _rows(g):
        d = len(cols)
        syn_ok &= ~np.logical_xor.reduce(hard[:, cols], axis=1)
        t = (lam[:, cols] - ne[:, e0:e0 + d]).astype(np.float32)
        ne2[:, e0:e0 + d] = r16(cn_minsum_f32(t))
    acc = orig.copy()
    for m, cols, e0 in reversed(_rows(g)):        # foldr (+) orig: rows in descending order (Min.hs:101)
        acc[:, cols] = (ne2[:, e0:e0 + len(cols)] + acc[:, cols]).astype(np.float32)
    return ne2, r16(acc), syn_ok


def decode_minsum_f16_flood(g, llr, max_iters):
    """llr [F, N] float32 -> bits [F, N] u8, iters [F], converged [F], trace (list over turns of lam [F, N],
    holding for a finished frame its returned LLR vector, as ldpc_decode_trace does)"""
    llr = np.asarray(llr, np.float32)
    F = llr.shape[0]
    orig = r16(llr)
    lam = orig.copy()
    ne = np.zeros((F, g.E), np.float32)
    iters = np.zeros(F, np.int32)
    conv = np.zeros(F, bool)
    live = np.ones(F, bool)
    out = orig.copy()
    trace = []
    for n in range(max_iters + 1):
        trace.append(np.where(live[:, None], lam, np.float32(0)))
        ne2, lam2, ok = step_minsum_f16_flood(g, orig, lam, ne)
        fin = live & ok                           # Min.hs:75: syndrome zero -> lam
        out[fin] = lam[fin]; conv[fin] = True; iters[fin] = n
        live &= ~ok
        if n >= max_iters:                        # Min.hs:76: out of turns -> channel LLRs
            iters[live] = n
            break
        if not live.any():
            break
        lam = np.where(live[:, None], lam2, lam)
        ne = np.where(live[:, None], ne2, ne)
    return (out > 0).astype(np.uint8), iters, conv, trace
